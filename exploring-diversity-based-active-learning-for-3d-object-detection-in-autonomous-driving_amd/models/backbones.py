"""Sparse 3-D ResNet middle encoder (reference det3d/models/backbones/scn.py:54-97,316-392).

Module tree and parameter names/layouts match the reference + spconv 1.2.1 so its
checkpoints load: ``middle_conv{0..3}.<i>.weight [kz,ky,kx,Cin,Cout]``, SparseBasicBlock
``conv{1,2}.{weight,bias}`` / ``bn{1,2}.*``.  The forward pass is a list of fused HIP layers
(csrc/spconv.hip): conv + bias + BN(eval) (+residual) (+ReLU) per launch.
"""
import ctypes

import numpy as np
import torch
from torch import nn

from .. import lib
from .. import detector_ops as D
from ..detector_ops import MFMA_PAIRS, fold_bn
from ..selector_ops import _ptr, _stream
from .registry import BACKBONES


class _SpConvParams(nn.Module):
    """Parameter holder with spconv's layout ``weight [*k, Cin, Cout]``."""

    def __init__(self, cin, cout, ksize, stride=1, padding=0, bias=False, subm=False):
        super().__init__()
        k = tuple(ksize) if isinstance(ksize, (tuple, list)) else (ksize,) * 3
        s = tuple(stride) if isinstance(stride, (tuple, list)) else (stride,) * 3
        p = tuple(padding) if isinstance(padding, (tuple, list)) else (padding,) * 3
        self.in_channels, self.out_channels = cin, cout
        self.kernel_size, self.stride, self.padding, self.subm = k, s, p, subm
        self.weight = nn.Parameter(torch.empty(*k, cin, cout))
        if bias:
            self.bias = nn.Parameter(torch.zeros(cout))
        else:
            self.register_parameter("bias", None)
        fan = cin * int(np.prod(k))
        nn.init.uniform_(self.weight, -1.0 / fan ** 0.5, 1.0 / fan ** 0.5)


class SubMConv3d(_SpConvParams):
    def __init__(self, cin, cout, ksize, bias=True, indice_key=None):
        super().__init__(cin, cout, ksize, 1, 0, bias, subm=True)
        self.indice_key = indice_key


class SparseConv3d(_SpConvParams):
    def __init__(self, cin, cout, ksize, stride=1, padding=0, bias=True):
        super().__init__(cin, cout, ksize, stride, padding, bias, subm=False)


class SparseBasicBlock(nn.Module):
    """conv1-bn1-relu-conv2-bn2-(+identity)-relu; the convs carry a bias (scn.py:68-73)."""

    def __init__(self, inplanes, planes, indice_key=None):
        super().__init__()
        self.conv1 = SubMConv3d(inplanes, planes, 3, bias=True, indice_key=indice_key)
        self.bn1 = nn.BatchNorm1d(planes, eps=1e-3, momentum=0.01)
        self.relu = nn.ReLU()
        self.conv2 = SubMConv3d(planes, planes, 3, bias=True, indice_key=indice_key)
        self.bn2 = nn.BatchNorm1d(planes, eps=1e-3, momentum=0.01)


class SparseTensor:
    """Minimal stand-in for spconv.SparseConvTensor: features [N,C] f32, indices [N,4] i32."""

    def __init__(self, features, indices, spatial_shape, batch_size, pair_rows=False):
        # pair_rows: ``features`` is in the f16x3 kernels' pair-row format (csrc/sp_rows.h); converted on first access
        self._features, self._pair = features, pair_rows
        self.indices = indices
        self.spatial_shape, self.batch_size = list(spatial_shape), batch_size

    @property
    def features(self):
        if self._pair:
            self._features, self._pair = D.rows_convert(self._features, to_pair=False), False
        return self._features


def _bn(c):
    return nn.BatchNorm1d(c, eps=1e-3, momentum=0.01)


class _Level:
    """Dense index grid of one resolution level, kept in HBM across calls."""

    def __init__(self, shape, max_batch, device):
        self.D, self.H, self.W = [int(s) for s in shape]
        self.max_batch = max_batch
        self.grid = torch.full((max_batch * self.D * self.H * self.W,), -1, dtype=torch.int32,
                               device=device)


def _i3(v):
    return (ctypes.c_int * 3)(*[int(x) for x in v])


class _SparseEncoderBase(nn.Module):
    def _stages(self):
        raise NotImplementedError

    def _prepare(self, device):
        """Pack weights / fold BN once per device (eval only)."""
        if getattr(self, "_packed_dev", None) == (device, D.MATH, D.SPCONV, D.L0, tuple(sorted(D.R16_COUTS)),
                                                  tuple(sorted(D.BLK_PAIRS))):
            return
        plan = []
        # The input level's rows may be renumbered (raster order, csrc/spconv_l0.hip) only if they never leave the encoder:
        # a strided conv must come before the first stage output (true for every shipped encoder; a stage that ends on the
        # input level keeps the caller's row order, like spconv's SubMConv3d)
        first = list(self._stages()[0].children()) if len(self._stages()) else []
        self._raster_ok = any(isinstance(m_, _SpConvParams) and not m_.subm for m_ in first)
        for seq in self._stages():
            mods = list(seq.children())
            i = 0
            while i < len(mods):
                m = mods[i]
                if isinstance(m, _SpConvParams):
                    bn = mods[i + 1]
                    scale, shift = fold_bn(bn)
                    w, scale = self._pack(m, device, scale.to(device), self._raster_ok)
                    plan.append(dict(kind="subm" if m.subm else "down", mod=m, w=w,
                                     scale=scale, shift=shift.to(device), relu=True,
                                     residual=False))
                    i += 3  # conv, bn, relu
                elif isinstance(m, SparseBasicBlock):
                    for conv, bn, last in ((m.conv1, m.bn1, False), (m.conv2, m.bn2, True)):
                        scale, shift = fold_bn(bn)
                        if conv.bias is not None:   # (x + b) * s + t
                            shift = shift + conv.bias.detach().float().to(scale.device) * scale
                        w, scale = self._pack(conv, device, scale.to(device), self._raster_ok)
                        plan.append(dict(kind="subm", mod=conv, w=w,
                                         scale=scale, shift=shift.to(device), relu=True,
                                         residual=last, block_start=not last))
                    i += 1
                else:
                    i += 1
            plan.append(dict(kind="stage_end"))
        self._plan = plan
        self._packed_dev = (device, D.MATH, D.SPCONV, D.L0, tuple(sorted(D.R16_COUTS)), tuple(sorted(D.BLK_PAIRS)))
        self._levels = {}

    @staticmethod
    def _pad_cin(m):
        """Input channels the layer is run with: a narrow first layer (5 -> 16) is zero-padded to
        16 input channels so it runs on the matrix cores like every other layer (bf16x6 only)."""
        if D.sparse_math() in ("bf16x6", "f16x3") and m.in_channels < 16 and (16, m.out_channels) in MFMA_PAIRS:
            return 16
        return m.in_channels

    @staticmethod
    def _pack(m, device, scale, raster_ok=False):
        """[kz,ky,kx,Cin,Cout] -> [K,Cin,Cout] (VALU kernel) or [Cout,K,Cin] (MFMA kernels, split into
        bf16 / f16 planes for the split arithmetics).  Returns (weights, scale): the f16x3 split folds
        its weight exponent into the layer's BN scale."""
        w = m.weight.detach().reshape(-1, m.in_channels, m.out_channels).float()
        cin = _SparseEncoderBase._pad_cin(m)
        if cin != m.in_channels:
            w = torch.nn.functional.pad(w, (0, 0, 0, cin - m.in_channels))
        if (cin, m.out_channels) in MFMA_PAIRS:
            w = w.permute(2, 0, 1).contiguous().to(device)
            if D.sparse_math() == "f16x3":
                planes, scale = D.split_f16x3(w, scale)
                if raster_ok and D.sparse_r16(cin, m.out_channels, w.shape[1]):
                    return D.pack_r16_f16x3(planes), scale
                blk = m.subm and D.sparse_blk(cin, m.out_channels, w.shape[1])    # block-staged kernel: the same image
                return (D.pack_glds_f16x3(planes) if (blk or D.sparse_glds(cin, m.out_channels)) else planes), scale
            return (D.split_bf16x3(w) if D.sparse_math() == "bf16x6" else w), scale
        return w.contiguous().to(device), scale

    @staticmethod
    def _conv(m, feats, nbr, K, step, residual, out, n, st, tmask=None, trng=None, io=0, items=None, plan=None):
        """One fused sparse layer (conv + folded BN + optional residual + ReLU)."""
        res_ptr = None if residual is None else _ptr(residual)
        cin = feats.shape[-1]                    # == m.in_channels, or 16 for a zero-padded narrow first layer
        mfma_pair = (cin, m.out_channels) in MFMA_PAIRS
        if isinstance(step["w"], D.R16Packed):
            # level-0 layer on raster rows: item stream, LDS-resident weights (csrc/spconv_l0.hip)
            lib.call("al3d_sp_conv_r16_f16x3", _ptr(feats), _ptr(nbr), nbr.shape[1], _ptr(items[1]), _ptr(items[0]), K,
                     _ptr(step["w"].data), cin, m.out_channels, _ptr(step["scale"]), _ptr(step["shift"]), res_ptr, 1,
                     _ptr(out), n, io, D.R16_TPW, st)
            return
        if plan is not None:
            # f16x3 arithmetic, block-staged gather: the union of a chunk's neighbourhoods staged once (csrc/spconv_blk.hip)
            lib.call("al3d_sp_conv_blk_f16x3", _ptr(feats), _ptr(nbr), nbr.shape[1], _ptr(tmask), _ptr(plan.hdr),
                     _ptr(plan.rows), _ptr(plan.loc), K, _ptr(step["w"].data), cin, m.out_channels, _ptr(step["scale"]),
                     _ptr(step["shift"]), res_ptr, 1, _ptr(out), n, io, st)
            return
        if mfma_pair and isinstance(step["w"], D.GldsPacked) and trng is not None and m.subm and K == 27 and \
                D.sparse_rng(cin, m.out_channels):
            # f16x3 arithmetic, LDS-DMA range gather: one staged index range per (tile, kz, ky) serves three taps
            lib.call("al3d_sp_conv_rng_f16x3", _ptr(feats), _ptr(nbr), nbr.shape[1], _ptr(tmask), _ptr(trng), K,
                     _ptr(step["w"].data), cin, m.out_channels, _ptr(step["scale"]), _ptr(step["shift"]), res_ptr,
                     1, _ptr(out), n, io, st)
            return
        if mfma_pair and isinstance(step["w"], D.GldsPacked):
            # f16x3 arithmetic, LDS-DMA row gather (full-line fetches) + producer-wave weight slabs
            lib.call("al3d_sp_conv_glds_f16x3_io", _ptr(feats), _ptr(nbr), nbr.shape[1], _ptr(tmask), K,
                     _ptr(step["w"].data), cin, m.out_channels, _ptr(step["scale"]), _ptr(step["shift"]), res_ptr,
                     1, _ptr(out), n, io, st)
            return
        if mfma_pair and step["w"].dtype == torch.float16 and tmask is not None:
            # f16x3 arithmetic, software-pipelined register-gather wave kernel on the tiled rulebook
            lib.call("al3d_sp_conv_wave2_f16x3_tiles_io", _ptr(feats), _ptr(nbr), nbr.shape[1], _ptr(tmask), K,
                     _ptr(step["w"]), cin, m.out_channels, _ptr(step["scale"]), _ptr(step["shift"]), res_ptr, 1,
                     _ptr(out), n, io, st)
            return
        if io:
            raise lib.Al3dError("sparse encoder: pair rows reached a layer without a tiled f16x3 kernel")
        if mfma_pair and step["w"].dtype == torch.float16:
            fn = "al3d_sp_conv_wave2_f16x3"           # the same kernel on a plain table
        elif mfma_pair and step["w"].dtype == torch.bfloat16:
            # bf16x6 arithmetic.  Measured per channel pair on the real rulebooks
            # (tools/bench_splayers.py): the software-pipelined wave kernel wins everywhere;
            # AL3D_SPCONV=wave|tile selects the older structures (same results bit for bit).
            fn = {"wave": "al3d_sp_conv_wave_bf16x6", "tile": "al3d_sp_conv_bf16x6"}.get(
                D.SPCONV, "al3d_sp_conv_wave2_bf16x6")
        else:
            fn = "al3d_sp_conv_mfma_f32" if mfma_pair else "al3d_sp_conv_f32"
        lib.call(fn, _ptr(feats), _ptr(nbr), K, _ptr(step["w"]), cin, m.out_channels,
                 _ptr(step["scale"]), _ptr(step["shift"]), res_ptr, 1, _ptr(out), n, st)

    def _level(self, shape, batch, device):
        key = tuple(int(s) for s in shape)
        lv = self._levels.get(key)
        if lv is None or lv.max_batch < batch:
            lv = _Level(key, max(batch, 1), device)
            self._levels[key] = lv
        return lv

    @staticmethod
    def _out_shape(shape, k, s, p):
        return [(shape[d] + 2 * p[d] - (k[d] - 1) - 1) // s[d] + 1 for d in range(3)]

    def build_rulebook(self, coords, batch_size, spatial_shape, frame_rows_max=0):
        """All index work of one batch.  It depends on the voxel coordinates only -- level grids,
        output sites of the strided convs (one small D2H each), every layer's tap-major table -- so
        the sweep can run it for batch i+1 on a side stream while batch i is being convolved.
        Returns ``dict(steps=[...])`` with one entry per item of the layer plan."""
        if self.training:
            raise RuntimeError("al3d sparse encoder implements the eval() sweep only")
        dev = coords.device
        self._prepare(dev)
        st = _stream()
        coords = coords.to(torch.int32).contiguous()
        shape = [int(s) for s in spatial_shape]
        n = coords.shape[0]
        perm, raster_status = None, None
        if D.sparse_raster() and any(isinstance(s_.get("w"), D.R16Packed) for s_ in self._plan):
            # the input level's rows renumbered in raster order (b, z, y, x): the order of a level's rows is free inside
            # the encoder (example["coordinates"] keeps the reference's first-appearance order), and raster order makes
            # the neighbour sets of consecutive rows contiguous index ranges (csrc/spconv_l0.hip)
            # (frame_rows_max > 0: the caller promises frame-sorted rows with at most that many rows per frame)
            perm, coords = D.raster_perm(coords, batch_size, shape, frame_rows_max)
            raster_status = D.raster_perm.last_status if frame_rows_max > 0 else None
        lv = self._level(shape, batch_size, dev)
        lib.call("al3d_sp_scatter_index", _ptr(coords), n, batch_size, lv.D, lv.H, lv.W, _ptr(lv.grid),
                 1, st)
        used = [(lv, coords, n)]
        nbr, nbr_key, trng, items, plan = None, None, None, None, None
        steps = []
        for pi, step in enumerate(self._plan):
            if step["kind"] == "stage_end":
                steps.append(dict(coords=coords, shape=shape, n=n))
                continue
            m = step["mod"]
            K = int(np.prod(m.kernel_size))
            # f16x3 (both matrix-core kernels): pitched table + per-tile tap masks; other arithmetics: plain table
            tiled = isinstance(step["w"], (D.GldsPacked, D.R16Packed)) or (isinstance(step["w"], torch.Tensor) and
                                                                           step["w"].dtype == torch.float16)
            if step["kind"] == "subm":
                key = (id(lv), m.kernel_size, tiled)
                if nbr_key != key:
                    if tiled:
                        pitch = lib.load().al3d_sp_table_pitch(n)
                        nbr = torch.empty((K, pitch), dtype=torch.int32, device=dev)
                        tmask = torch.empty((pitch // 32,), dtype=torch.int32, device=dev)
                        lib.call("al3d_sp_subm_table_tiles", _ptr(coords), n, batch_size, lv.D, lv.H, lv.W,
                                 _ptr(lv.grid), *m.kernel_size, _ptr(nbr), pitch, _ptr(tmask), st)
                    else:
                        nbr, tmask = torch.empty((K, max(n, 1)), dtype=torch.int32, device=dev), None
                        lib.call("al3d_sp_subm_table", _ptr(coords), n, batch_size, lv.D, lv.H, lv.W,
                                 _ptr(lv.grid), *m.kernel_size, _ptr(nbr), st)
                    nbr_key = key
                    trng, items, plan = None, None, None
                blk = tiled and m.subm and isinstance(step["w"], D.GldsPacked) and D.sparse_blk(self._pad_cin(m), m.out_channels, K)
                if blk and plan is None:
                    plan = D.block_plan(nbr, n, self._pad_cin(m), m.out_channels)     # once per table, shared by the level
                if tiled and not blk and trng is None and K == 27 and D.sparse_rng(self._pad_cin(m), m.out_channels):
                    # (lo, len) of every (tile, kz, ky) group: once per table, shared by the level's layers
                    trng = torch.empty((max(nbr.shape[1] // 32, 1), 9, 2), dtype=torch.int32, device=dev)
                    lib.call("al3d_sp_tile_ranges", _ptr(nbr), nbr.shape[1], K, n, _ptr(trng), st)
                if isinstance(step["w"], D.R16Packed) and items is None:
                    items = D.tile_items(nbr, n, tmask)          # once per table, shared by the level's layers
                steps.append(dict(nbr=nbr, n=n, K=K, tmask=tmask, trng=None if blk else trng, plan=plan if blk else None,
                                  items=items if isinstance(step["w"], D.R16Packed) else None))
            else:
                oshape = self._out_shape(shape, m.kernel_size, m.stride, m.padding)
                olv = self._level(oshape, batch_size, dev)
                cap = min(n * K, batch_size * olv.D * olv.H * olv.W)
                ocoords = torch.empty((max(cap, 1), 4), dtype=torch.int32, device=dev)
                counter = torch.zeros(1, dtype=torch.int32, device=dev)
                ks, ss, ps = _i3(m.kernel_size), _i3(m.stride), _i3(m.padding)
                # the order of the new level's rows: column by column when its layers run on the block-staged kernel
                # (csrc/spconv_blk.hip), raster otherwise
                nxt = next((s_ for s_ in self._plan[pi + 1:] if s_["kind"] != "stage_end"), None)
                blocked = nxt is not None and nxt["kind"] == "subm" and \
                    D.sparse_blk_order(self._pad_cin(nxt["mod"]), nxt["mod"].out_channels, int(np.prod(nxt["mod"].kernel_size)))
                sites = "al3d_sp_down_sites_blocked" if blocked else "al3d_sp_down_sites"
                ws = torch.empty(getattr(lib.load(), sites + "_workspace_bytes")(batch_size, olv.D, olv.H, olv.W),
                                 dtype=torch.uint8, device=dev)
                lib.call(sites, _ptr(coords), n, ks, ss, ps, batch_size, olv.D, olv.H,
                         olv.W, _ptr(olv.grid), _ptr(ocoords), _ptr(counter), cap, _ptr(ws), st)
                n_out = int(counter.item())      # one small D2H per stage
                if raster_status is not None:    # the level-0 order's promise, checked where the stream is synchronised anyway
                    D.check_raster_status(raster_status)
                    raster_status = None
                ocoords = ocoords[:n_out]
                if not blocked and n_out > 0 and nxt is not None and nxt["kind"] == "subm" and \
                        int(np.prod(nxt["mod"].kernel_size)) == 27 and nxt["mod"].out_channels in D.MASK_SORT:
                    # rows of the new level grouped by tap mask inside windows of raster rows: more whole-tile tap skips
                    sorted_coords = torch.empty_like(ocoords)
                    msw = torch.empty(lib.load().al3d_sp_mask_window_sort_workspace_bytes(n_out), dtype=torch.uint8, device=dev)
                    lib.call("al3d_sp_mask_window_sort", _ptr(ocoords), n_out, batch_size, olv.D, olv.H, olv.W, _ptr(olv.grid),
                             D.MASK_SORT_WINDOWS.get(nxt["mod"].out_channels, D.MASK_SORT_WINDOW), _ptr(sorted_coords), _ptr(msw), st)
                    ocoords = sorted_coords
                used.append((olv, ocoords, n_out))
                if tiled:
                    pitch = lib.load().al3d_sp_table_pitch(n_out)
                    dnbr = torch.empty((K, pitch), dtype=torch.int32, device=dev)
                    dmask = torch.empty((pitch // 32,), dtype=torch.int32, device=dev)
                    lib.call("al3d_sp_down_table_tiles", _ptr(ocoords), n_out, ks, ss, ps, batch_size, lv.D, lv.H,
                             lv.W, _ptr(lv.grid), _ptr(dnbr), pitch, _ptr(dmask), st)
                else:
                    dnbr, dmask = torch.empty((K, max(n_out, 1)), dtype=torch.int32, device=dev), None
                    lib.call("al3d_sp_down_table", _ptr(ocoords), n_out, ks, ss, ps, batch_size, lv.D, lv.H,
                             lv.W, _ptr(lv.grid), _ptr(dnbr), st)
                steps.append(dict(nbr=dnbr, n=n_out, K=K, tmask=dmask,
                                  items=D.tile_items(dnbr, n_out, dmask) if isinstance(step["w"], D.R16Packed) else None))
                coords, n, shape, lv = ocoords, n_out, oshape, olv
                nbr_key = None
        if raster_status is not None:
            D.check_raster_status(raster_status)
        for g, c, cnt in used:      # leave every level grid clean for the next call
            lib.call("al3d_sp_scatter_index", _ptr(c), cnt, batch_size, g.D, g.H, g.W, _ptr(g.grid), 0, st)
        # the zero-filled dense BEV buffer the last stage scatters into (537 MB at batch 32): also
        # coordinate-independent work that can be done ahead
        last_c = [st_["mod"].out_channels for st_ in self._plan if st_["kind"] != "stage_end"][-1]
        dense = torch.zeros((batch_size, shape[1], shape[2], last_c * shape[0]), dtype=torch.float32, device=dev)
        return dict(steps=steps, batch_size=batch_size, dense=dense, perm=perm)

    def _run(self, feats, coords, batch_size, spatial_shape, book=None):
        """Returns (final SparseTensor, [SparseTensor per stage]).  ``book``: a rulebook built earlier
        by ``build_rulebook`` for these coordinates (else it is built here)."""
        if book is None:
            book = self.build_rulebook(coords, batch_size, spatial_shape)
        dev = feats.device
        st = _stream()
        feats = feats.float().contiguous()
        middle = []
        identity, identity_pair = None, False
        # f16x3: between two layers that both run a tiled matrix-core kernel the rows travel as pair rows (the
        # producer's epilogue splits once; csrc/sp_rows.h).  The first layer reads the VFE's f32 rows, the last one
        # writes f32 rows for the dense scatter.
        def pairable(step_, b_):
            return (D.SPROWS == "pair" and step_["kind"] != "stage_end" and b_.get("tmask") is not None and
                    (self._pad_cin(step_["mod"]), step_["mod"].out_channels) in MFMA_PAIRS and
                    (isinstance(step_["w"], (D.GldsPacked, D.R16Packed)) or (isinstance(step_["w"], torch.Tensor) and
                                                                             step_["w"].dtype == torch.float16)))
        convs = [(s_, b_) for s_, b_ in zip(self._plan, book["steps"]) if s_["kind"] != "stage_end"]
        ci, pair = 0, False
        perm = book.get("perm")
        for step, b in zip(self._plan, book["steps"]):
            if step["kind"] == "stage_end":
                middle.append(SparseTensor(feats, b["coords"], b["shape"], batch_size, pair_rows=pair))
                hook = getattr(self, "stage_hook", None)        # the sweep's pipeline: "stage k of the encoder is enqueued"
                if hook is not None:
                    hook(len(middle) - 1)
                continue
            m = step["mod"]
            if perm is not None or feats.shape[-1] != self._pad_cin(m):
                assert not pair
                # the voxel features in the encoder's row order, zero-padded to the first layer's input width
                # (as pair rows when the first layer is an item-stream layer: AL3D_L0_ROWS)
                first_pair = D.L0_ROWS == "pair" and isinstance(step["w"], D.R16Packed) and pairable(step, b)
                feats = D.rows_gather_pad(feats, perm, self._pad_cin(m), to_pair=first_pair)
                perm, pair = None, first_pair
            if step.get("block_start"):
                identity, identity_pair = feats, pair
            ok = pairable(step, b)
            assert ok or not pair, "pair rows reached a layer that cannot read them"
            # 16-channel rows stay f32: on the level-0 layers the pair-row epilogue costs more (+12 %) than the
            # consumers gain (-3 %); from 32 channels on the consumers are the LDS-DMA kernels (-5..10 %)
            # (the item-stream kernels of level 0 are bound by instruction issue, the split is a third of their vector
            # instructions: there 16-channel rows travel as pair rows too)
            out_pair = ok and ci + 1 < len(convs) and pairable(*convs[ci + 1]) and (
                m.out_channels >= 32 or (D.L0_ROWS == "pair" and isinstance(step["w"], D.R16Packed) and
                                         isinstance(convs[ci + 1][0]["w"], D.R16Packed)))
            res = identity if step.get("residual") else None
            io = ((D.IO_IN_PAIR if pair else 0) | (D.IO_OUT_PAIR if out_pair else 0) |
                  (D.IO_RES_PAIR if (res is not None and identity_pair) else 0))
            out = torch.empty((b["n"], m.out_channels), dtype=torch.float32, device=dev)
            self._conv(m, feats, b["nbr"], b["K"], step, res, out, b["n"], st,
                       tmask=b.get("tmask"), trng=b.get("trng"), io=io, items=b.get("items"), plan=b.get("plan"))
            feats, pair = out, out_pair
            ci += 1
        assert not pair
        last = middle[-1]
        return SparseTensor(feats, last.indices, last.spatial_shape, batch_size), middle

    @staticmethod
    def dense_nhwc(sp, out=None):
        """``ret.dense()`` + ``view(N, C*D, H, W)`` in NHWC: [B, H, W, C*D], channel = c*D + z.
        ``out``: an already zero-filled buffer of that shape (the rulebook pass prepares one)."""
        D, H, W = sp.spatial_shape
        C = sp.features.shape[1]
        if out is None or tuple(out.shape) != (sp.batch_size, H, W, C * D):
            out = torch.zeros((sp.batch_size, H, W, C * D), dtype=torch.float32, device=sp.features.device)
        lib.call("al3d_sp_to_dense_nhwc", _ptr(sp.features), _ptr(sp.indices), sp.features.shape[0], C,
                 sp.batch_size, D, H, W, _ptr(out), _stream())
        return out


@BACKBONES.register_module
class FPNSpMiddleResNetFHD(_SparseEncoderBase):
    def __init__(self, num_input_features=128, norm_cfg=None, name="SpMiddleResNetFHD", **kwargs):
        super().__init__()
        self.name = name
        self.middle_conv0 = nn.Sequential(
            SubMConv3d(num_input_features, 16, 3, bias=False, indice_key="res0"), _bn(16), nn.ReLU(),
            SparseBasicBlock(16, 16, indice_key="res0"), SparseBasicBlock(16, 16, indice_key="res0"),
            SparseConv3d(16, 32, 3, 2, padding=1, bias=False), _bn(32), nn.ReLU())
        self.middle_conv1 = nn.Sequential(
            SparseBasicBlock(32, 32, indice_key="res1"), SparseBasicBlock(32, 32, indice_key="res1"),
            SparseConv3d(32, 64, 3, 2, padding=1, bias=False), _bn(64), nn.ReLU())
        self.middle_conv2 = nn.Sequential(
            SparseBasicBlock(64, 64, indice_key="res2"), SparseBasicBlock(64, 64, indice_key="res2"),
            SparseConv3d(64, 128, 3, 2, padding=[0, 1, 1], bias=False), _bn(128), nn.ReLU())
        self.middle_conv3 = nn.Sequential(
            SparseBasicBlock(128, 128, indice_key="res3"), SparseBasicBlock(128, 128, indice_key="res3"),
            SparseConv3d(128, 128, (3, 1, 1), (2, 1, 1), bias=False), _bn(128), nn.ReLU())

    def _stages(self):
        return [self.middle_conv0, self.middle_conv1, self.middle_conv2, self.middle_conv3]

    def rulebook_for(self, coors, batch_size, input_shape, frame_rows_max=0):
        return self.build_rulebook(coors, batch_size, np.array(input_shape[::-1]) + [1, 0, 0], frame_rows_max)

    def forward(self, voxel_features, coors, batch_size, input_shape, book=None, frame_rows_max=0):
        """-> (dense NHWC [B,128,128,256], middle list of 4 SparseTensor) -- the reference
        returns NCHW (scn.py:371-392); this build keeps activations channels-last."""
        sparse_shape = np.array(input_shape[::-1]) + [1, 0, 0]
        if book is None:
            book = self.build_rulebook(coors, batch_size, sparse_shape, frame_rows_max)
        final, middle = self._run(voxel_features, coors, batch_size, sparse_shape, book=book)
        return self.dense_nhwc(final, out=book.pop("dense", None)), middle
