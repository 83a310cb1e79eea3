"""CPU suite: the native sweep-file reader pool (csrc/reader.cpp) -- host code, no GPU needed.
Mirrors read_file's rule (det3d/datasets/pipelines/loading.py:17-24): whole 5-float rows only."""
import ctypes
import os

import numpy as np
import pytest


def _files(tmp_path, sizes):
    rng = np.random.default_rng(3)
    paths, arrays = [], []
    for i, n in enumerate(sizes):
        a = rng.normal(size=n).astype(np.float32)           # n floats (not necessarily a multiple of 5)
        p = str(tmp_path / f"f{i}.bin")
        a.tofile(p)
        paths.append(p)
        arrays.append(a)
    return paths, arrays


def test_reader_reads_whole_rows_in_parallel(tmp_path):
    from al3d.datasets import SweepFileReader
    sizes = [5 * 1000, 5 * 17 + 3, 0, 5 * 40000 + 4, 5, 4] + [5 * (100 + 7 * i) for i in range(40)]
    paths, arrays = _files(tmp_path, sizes)
    rd = SweepFileReader(threads=6)
    rows = rd.plan(paths)
    assert rows.tolist() == [n // 5 for n in sizes]
    off = np.concatenate([[0], np.cumsum(rows)])
    buf = np.full(int(off[-1]) * 5 + 7, np.float32(-77.0), dtype=np.float32)
    jobs = [rd.submit(paths, off[:-1], rows, buf.ctypes.data, int(off[-1]) * 20) for _ in range(3)]   # re-reads are idempotent
    for j in jobs:
        rd.wait(j)
    for a, o, r in zip(arrays, off[:-1], rows):
        assert np.array_equal(buf[5 * o:5 * (o + r)].view(np.int32), a[:5 * r].view(np.int32))
    assert (buf[int(off[-1]) * 5:] == np.float32(-77.0)).all()          # nothing written past the planned rows
    rd.close()


def test_reader_errors_surface_on_the_waiting_thread(tmp_path):
    from al3d import lib
    from al3d.datasets import SweepFileReader
    paths, _ = _files(tmp_path, [50, 50])
    rd = SweepFileReader(threads=2)
    with pytest.raises(lib.Al3dError, match="nope.bin"):
        rd.plan(paths + [str(tmp_path / "nope.bin")])
    rows = rd.plan(paths)
    buf = np.zeros(100, dtype=np.float32)
    with pytest.raises(lib.Al3dError, match="does not fit"):
        rd.submit(paths, [0, 10], rows, buf.ctypes.data, 10 * 20)       # second file would overrun the buffer
    os.remove(paths[1])                                                 # vanishes between plan and read
    job = rd.submit(paths, [0, 10], rows, buf.ctypes.data, 400)
    with pytest.raises(lib.Al3dError, match="open"):
        rd.wait(job)
    with pytest.raises(lib.Al3dError, match="unknown job"):
        rd.wait(job)                                                    # a job is forgotten once waited for
    rd.close()
