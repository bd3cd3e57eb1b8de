"""Writer processes for the per-sample .npz files (src/training/callbacks.py:17-44 layout).

``python -m mile_amd.sample_writer`` is a worker: it reads length-prefixed pickled tasks from stdin and
writes ``<base>/<chain>/sample_<n>.npz`` files.  Workers are plain subprocesses (no multiprocessing:
nothing of the parent's ``__main__`` is re-imported), import numpy only and run with the GPUs hidden, so
they can never open the device.  ``WriterPool`` feeds them from threads so the stepping loop never blocks.
"""
from __future__ import annotations

import os
import pickle
import queue
import struct
import subprocess
import sys
import threading
import zlib
from pathlib import Path

import numpy as np


# ---- the .npz container, written directly --------------------------------------------------------------------------
# np.savez_compressed (what the reference calls, callbacks.py:42) = a zip archive with one deflated `<key>.npy` member per
# array.  fp32 posterior samples do not compress (measured on a B2 sample: 35 336 -> 34 637 bytes, 2 %) while zlib spends
# 0.85 ms per 35 KB file on them at ANY level >= 1 -- 128 000 files of the stock schedule = 109 core-seconds, 3-4x the
# stepping time of the whole sampling phase on the 16 host cores of a GPU.  The files here are the same archives -- same
# member names, same .npy headers, method 8 (deflate), readable by np.load / zipfile exactly like numpy's own -- whose
# deflate streams are written at MILE_NPZ_DEFLATE_LEVEL (default 0: stored blocks, 2 % larger files, 0.05 ms instead of
# 0.85 ms of zlib per file; 6 = numpy's level).
_NPY_HEADERS: dict = {}


def _npy_header(dtype: np.dtype, shape: tuple) -> bytes:
    key = (dtype.str, tuple(shape))
    h = _NPY_HEADERS.get(key)
    if h is None:
        import io
        from numpy.lib import format as fmt
        bio = io.BytesIO()
        fmt.write_array_header_1_0(bio, {'descr': fmt.dtype_to_descr(dtype), 'fortran_order': False, 'shape': tuple(shape)})
        h = _NPY_HEADERS[key] = bio.getvalue()
    return h


def deflate_level() -> int:
    try:
        return min(9, max(0, int(os.environ.get('MILE_NPZ_DEFLATE_LEVEL', '0'))))
    except ValueError:
        return 0


def write_npz(path, members, level: int | None = None):
    """members: [(key, C-contiguous ndarray)] -> a zip archive np.load reads as np.savez_compressed's."""
    level = deflate_level() if level is None else level
    parts, central, offset = [], [], 0
    for key, a in members:
        a = np.asarray(a)
        if not a.flags.c_contiguous:              # (np.ascontiguousarray would turn a 0-d array into shape (1,))
            a = np.array(a, order='C')
        data = _npy_header(a.dtype, a.shape) + a.tobytes()
        crc = zlib.crc32(data)
        co = zlib.compressobj(level, zlib.DEFLATED, -15)
        comp = co.compress(data) + co.flush()
        name = (key + '.npy').encode()
        # local file header / central directory entry (PKZIP appnote 4.3.7, 4.3.12): version 2.0, no flags, method 8
        # (deflate), DOS time 00:00 / date 1980-01-01, sizes < 4 GiB (no zip64 extra field needed)
        parts.append(struct.pack('<IHHHHHIIIHH', 0x04034b50, 20, 0, 8, 0, 0x21, crc, len(comp), len(data), len(name), 0) + name)
        parts.append(comp)
        central.append(struct.pack('<IHHHHHHIIIHHHHHII', 0x02014b50, 20, 20, 0, 8, 0, 0x21, crc, len(comp), len(data), len(name),
                                   0, 0, 0, 0, 0o600 << 16, offset) + name)
        offset += len(parts[-2]) + len(comp)
    cd = b''.join(central)
    end = struct.pack('<IHHHHIIH', 0x06054b50, 0, 0, len(central), len(central), len(cd), offset, 0)
    with open(path, 'wb') as f:
        f.write(b''.join(parts) + cd + end)


def write_chain_samples(leaves, rows: np.ndarray, base: str, idx: int, ns) -> int:
    path = Path(base) / f'{int(idx)}'
    path.mkdir(parents=True, exist_ok=True)
    level = deflate_level()
    for row, n in zip(rows, ns):
        write_npz(path / f'sample_{int(n)}.npz',
                  [(name, row[off:off + int(np.prod(shape))].reshape(shape)) for name, off, shape in leaves], level)
    return len(ns)


def _worker_main():
    inp, out = sys.stdin.buffer, sys.stdout.buffer
    done = 0
    while True:
        hdr = inp.read(8)
        if len(hdr) < 8:
            break
        (n,) = struct.unpack('<q', hdr)
        if n == 0:
            break
        task = pickle.loads(inp.read(n))
        done += write_chain_samples(*task)
    out.write(struct.pack('<q', done))
    out.flush()


class WriterPool:
    def __init__(self, n_workers: int):
        env = dict(os.environ, HIP_VISIBLE_DEVICES='', CUDA_VISIBLE_DEVICES='', ROCR_VISIBLE_DEVICES='')
        root = str(Path(__file__).resolve().parents[1])
        env['PYTHONPATH'] = root + os.pathsep + env.get('PYTHONPATH', '')
        self.procs, self.queues, self.threads = [], [], []
        for _ in range(max(1, n_workers)):
            p = subprocess.Popen([sys.executable, '-m', 'mile_amd.sample_writer'], stdin=subprocess.PIPE,
                                 stdout=subprocess.PIPE, env=env)
            q: queue.Queue = queue.Queue()
            t = threading.Thread(target=self._feed, args=(p, q), daemon=True)
            t.start()
            self.procs.append(p)
            self.queues.append(q)
            self.threads.append(t)
        self._next = 0
        self.submitted = 0

    @staticmethod
    def _feed(p, q):
        while True:
            task = q.get()
            if task is None:
                p.stdin.write(struct.pack('<q', 0))
                p.stdin.flush()
                return
            blob = pickle.dumps(task, protocol=pickle.HIGHEST_PROTOCOL)
            p.stdin.write(struct.pack('<q', len(blob)))
            p.stdin.write(blob)
            p.stdin.flush()

    def submit(self, leaves, rows: np.ndarray, base: str, idx: int, ns):
        self.queues[self._next % len(self.queues)].put((leaves, rows, base, idx, list(ns)))
        self._next += 1
        self.submitted += len(ns)

    def close(self) -> int:
        """Flush all tasks; returns the number of files written (raises if a worker failed)."""
        for q in self.queues:
            q.put(None)
        for t in self.threads:
            t.join()
        total = 0
        for p in self.procs:
            data = p.stdout.read(8)
            rc = p.wait()
            if rc != 0 or len(data) < 8:
                raise RuntimeError(f'sample writer process failed (exit code {rc})')
            total += struct.unpack('<q', data)[0]
        if total != self.submitted:
            raise RuntimeError(f'sample writers wrote {total} of {self.submitted} files')
        return total


if __name__ == '__main__':
    _worker_main()
