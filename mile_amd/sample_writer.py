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
from pathlib import Path

import numpy as np


def write_chain_samples(leaves, rows: np.ndarray, base: str, idx: int, ns) -> int:
    path = Path(base) / f'{int(idx)}'
    path.mkdir(parents=True, exist_ok=True)
    for row, n in zip(rows, ns):
        np.savez_compressed(path / f'sample_{int(n)}.npz',
                            **{name: row[off:off + int(np.prod(shape))].reshape(shape) for name, off, shape in leaves})
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
