"""Streaming sweep loader: real (or synthetic on-disk) nuScenes frames straight from their ``.bin`` files.

The reference overlaps 8 DataLoader worker processes with the GPU (det3d/datasets/loader/build_loader.py:23-59):
each worker reads the key frame + 9 sweeps of a sample, filters / transforms / concatenates them with numpy
(det3d/datasets/pipelines/loading.py:17-63,98-126) and pickles the result to the trainer.  Here

* a native reader pool (csrc/reader.cpp, ``al3d_reader_*``) preads the files of a whole BATCH into one pinned
  staging buffer, ``depth`` batches ahead of the GPU,
* one asynchronous H2D copy per batch moves the raw bytes,
* ``al3d_merge_sweeps_batch_f32`` does remove_close / float64 transform / time column / compaction for all frames of
  the batch at once (bit-identical per frame to the single-frame kernel and to the reference loader's golden),
* the device voxelizer turns the merged clouds into the ``example`` dict the detector reads,

so under the sweep's two-stream pipeline (``al3d/sweep.py``) file I/O, upload, merge and voxelization of batch i+1 all
run while batch i is convolved.  ``FileSweepLoader`` is a drop-in for ``DeviceSweepLoader`` (same example keys).
"""
import ctypes
import os

import numpy as np
import torch

from .. import lib
from ..detector_ops import Voxelizer


def usable_cores():
    """Cores this process may really use: affinity mask capped by the cgroup CPU quota."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = max(1, min(n, int(float(quota) / float(period))))
    except Exception:
        pass
    return n


class SweepFileReader:
    """ctypes face of the native reader pool."""

    def __init__(self, threads=8):
        self._h = ctypes.c_void_p()
        lib.call("al3d_reader_create", int(threads), ctypes.byref(self._h))

    def close(self):
        if self._h:
            lib.load().al3d_reader_destroy(self._h)
            self._h = ctypes.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @staticmethod
    def _paths(paths):
        return (ctypes.c_char_p * len(paths))(*[os.fsencode(p) for p in paths])

    def plan(self, paths):
        """-> int64 rows per file (whole 20-byte records; the reference drops a trailing partial row)."""
        rows = np.zeros(len(paths), dtype=np.int64)
        total = lib.load().al3d_reader_plan(self._paths(paths), len(paths), rows.ctypes.data_as(ctypes.c_void_p))
        if total < 0:
            lib.check(int(total), "al3d_reader_plan")
        return rows

    def submit(self, paths, row_off, rows, dst_ptr, dst_bytes):
        row_off = np.ascontiguousarray(row_off, dtype=np.int64)
        rows = np.ascontiguousarray(rows, dtype=np.int64)
        job = lib.load().al3d_reader_submit(self._h, self._paths(paths), len(paths),
                                            row_off.ctypes.data_as(ctypes.c_void_p),
                                            rows.ctypes.data_as(ctypes.c_void_p), ctypes.c_void_p(dst_ptr),
                                            int(dst_bytes))
        if job < 0:
            lib.check(int(job), "al3d_reader_submit")
        return int(job)

    def wait(self, job):
        lib.call("al3d_reader_wait", self._h, int(job))


class _Staged:
    """One batch in flight on the host side: its pinned buffer, layout and read job."""
    __slots__ = ("pinned", "job", "total", "nf", "B", "o_off", "o_xf", "o_tl", "o_has", "o_key", "o_ff", "nbytes",
                 "ids", "slot", "extra")                     # extra: a subclass's own per-batch state (camera decode jobs)


class FileSweepLoader:
    """Batches of ``example`` dicts (voxelnet.py:84-97 keys) read from the frames' ``.bin`` files.

    ``infos``: nuScenes info dicts (``lidar_path``, ``sweeps[i]{lidar_path, transform_matrix, time_lag}``,
    nusc_common.py:410-419).  Sweep order = list order (the reference draws a fresh random order per sample and is
    therefore not reproducible)."""

    def __init__(self, infos, voxel_cfg, anchors, batch_size=8, device="cuda", nsweeps=10, root=None, threads=8,
                 indices=None, depth=2, min_distance=1.0):
        self.infos = infos
        self.batch_size = int(batch_size)
        self.device = torch.device(device)
        self.nsweeps, self.root, self.min_distance = int(nsweeps), root, float(min_distance)
        self.indices = list(range(len(infos))) if indices is None else list(indices)
        self.depth = max(1, int(depth))
        self.reader = SweepFileReader(threads)
        self.voxelizer = Voxelizer(voxel_cfg["range"], voxel_cfg["voxel_size"], voxel_cfg["max_points_in_voxel"],
                                   voxel_cfg["max_voxel_num"], max_batch=self.batch_size, device=self.device)
        self.anchors = [torch.as_tensor(a, dtype=torch.float32, device=self.device) for a in (anchors or [])]
        self.dataset = infos
        self.sampler = self.indices
        # ring of depth + 1 pinned buffers: the one being uploaded is never the one being refilled
        self._pinned = [None] * (self.depth + 1)
        self._upload_done = [None] * (self.depth + 1)
        self.bytes_read = 0

    def __len__(self):
        return (len(self.indices) + self.batch_size - 1) // self.batch_size

    def _path(self, p):
        p = str(p)
        return p if self.root is None or os.path.isabs(p) else os.path.join(self.root, p)

    # transform rule of al3d_merge_sweeps_batch_rule_f32: 0 = det3d's loader (one float64 4 x 4 product, rounded once)
    rule = 0

    def _frame_files(self, info):
        """One frame's files in merge order: (paths, 3x4 / 4x4 float64 transforms or None, time lags, key-frame flags).
        det3d infos (nusc_common.py:410-419): ``lidar_path``, ``sweeps[i]{lidar_path, transform_matrix, time_lag}``."""
        assert self.nsweeps - 1 <= len(info["sweeps"]), \
            f"nsweeps {self.nsweeps} should not greater than list length {len(info['sweeps'])}."
        paths, xforms, lags, keys = [self._path(info["lidar_path"])], [None], [0.0], [1]
        for k in range(self.nsweeps - 1):
            sw = info["sweeps"][k]
            paths.append(self._path(sw["lidar_path"]))
            xforms.append(sw.get("transform_matrix"))
            lags.append(float(sw["time_lag"]))
            keys.append(0)
        return paths, xforms, lags, keys

    def _start(self, b):
        """Plan batch b, lay out its staging buffer, start the reads (returns immediately)."""
        ids = self.indices[b * self.batch_size:(b + 1) * self.batch_size]
        paths, xforms, lags, keys, first = [], [], [], [], [0]
        for i in ids:
            p_, x_, l_, k_ = self._frame_files(self.infos[i])
            paths += p_
            xforms += x_
            lags += l_
            keys += k_
            first.append(len(paths))
        rows = self.reader.plan(paths)
        st = _Staged()
        st.ids, st.nf, st.B = ids, len(paths), len(ids)
        st.total = int(rows.sum())
        o = st.total * 20
        o += (-o) % 8
        st.o_off = o
        o += 8 * (st.nf + 1)
        st.o_xf = o
        o += 96 * st.nf
        st.o_tl = o
        o += 8 * st.nf
        st.o_has = o
        o += st.nf
        st.o_key = o
        o += st.nf
        o += (-o) % 4
        st.o_ff = o
        o += 4 * (st.B + 1)
        st.nbytes = o
        st.slot = b % len(self._pinned)
        if self._upload_done[st.slot] is not None:       # the buffer's previous upload must have left the host
            self._upload_done[st.slot].synchronize()
            self._upload_done[st.slot] = None
        if self._pinned[st.slot] is None or self._pinned[st.slot].numel() < st.nbytes:
            self._pinned[st.slot] = torch.empty(int(st.nbytes * 1.25) + 4096, dtype=torch.uint8).pin_memory()
        st.pinned = self._pinned[st.slot]
        host = st.pinned.numpy()
        off = host[st.o_off:st.o_xf].view(np.int64)
        off[0] = 0
        np.cumsum(rows, out=off[1:])
        xf = host[st.o_xf:st.o_tl].view(np.float64).reshape(st.nf, 12)
        has = host[st.o_has:st.o_has + st.nf]
        for i, t in enumerate(xforms):
            has[i] = 0 if t is None else 1
            xf[i] = 0.0 if t is None else np.asarray(t, dtype=np.float64)[:3, :].reshape(12)
        host[st.o_tl:st.o_has].view(np.float64)[:] = lags
        host[st.o_key:st.o_key + st.nf] = keys
        host[st.o_ff:st.o_ff + 4 * (st.B + 1)].view(np.int32)[:] = first
        st.job = self.reader.submit(paths, off[:-1], rows, st.pinned.data_ptr(), st.total * 20)
        self.bytes_read += st.total * 20
        return st

    def _finish(self, st):
        """Wait for the reads of a staged batch, upload, merge, voxelize -> example dict."""
        self.reader.wait(st.job)
        dev = self.device
        stream = torch.cuda.current_stream(dev)
        buf = torch.empty(max(st.nbytes, 1), dtype=torch.uint8, device=dev)
        buf[:st.nbytes].copy_(st.pinned[:st.nbytes], non_blocking=True)
        ev = torch.cuda.Event()
        ev.record(stream)
        self._upload_done[st.slot] = ev
        out = torch.empty((max(st.total, 1), 5), dtype=torch.float32, device=dev)
        frame_off = torch.empty((st.B + 1,), dtype=torch.int64, device=dev)
        ws = torch.empty(lib.load().al3d_merge_sweeps_workspace_bytes(st.total), dtype=torch.uint8, device=dev)
        base = buf.data_ptr()
        rg = getattr(self, "point_range", None)              # PointsRangeFilter of the BEVFusion test pipeline (else None)
        rg = None if rg is None else (ctypes.c_float * 6)(*[float(np.float32(v)) for v in rg])
        lib.call("al3d_merge_sweeps_batch_range_f32", base, base + st.o_off, st.nf, st.total, base + st.o_xf, base + st.o_has,
                 base + st.o_tl, base + st.o_key, base + st.o_ff, st.B, self.min_distance, int(self.rule), rg, out.data_ptr(),
                 frame_off.data_ptr(), ws.data_ptr(), stream.cuda_stream)
        v = self.voxelizer(out, frame_off)
        gs = self.voxelizer.grid_size
        return {
            "voxel_features": v["feat"], "coordinates": v["coords"], "num_points": v["num_points"],
            "num_voxels": v["num_voxels"], "voxel_cap": v["voxel_cap"],
            "shape": np.tile(np.asarray(gs, dtype=np.int64)[None], (st.B, 1)),
            "anchors": self.anchors,
            "metadata": [{"token": str(self.infos[i].get("token", f"frame{i:06d}")), "index": i} for i in st.ids],
            "points": out, "point_offsets": frame_off,
        }

    def __iter__(self):
        nb = len(self)
        staged = {}
        try:
            for b in range(min(self.depth, nb)):
                staged[b] = self._start(b)
            for b in range(nb):
                st = staged.pop(b)
                ex = self._finish(st)
                if b + self.depth < nb:
                    staged[b + self.depth] = self._start(b + self.depth)   # read ahead while batch b runs downstream
                yield ex
        finally:
            # abandoned midway (exception downstream, break, a second iter()): the reads already submitted still
            # write into pinned slots the next iteration will reuse -- wait them out before anyone can
            for st in staged.values():
                try:
                    self.reader.wait(st.job)
                except Exception:
                    pass
                self._abandon(st)
            staged.clear()

    def _abandon(self, st):
        """Hook: a staged batch that will never be finished (subclasses drop what they attached to it)."""
