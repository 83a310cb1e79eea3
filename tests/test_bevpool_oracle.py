"""CPU suite: the BEV-pooling oracle (oracle/al3d_oracle_detector.c: al3d_oracle_bev_pool) against an independent
torch statement of bevfusion/mmdet3d/models/vtransforms/base.py:127-163 + ops/bev_pool (index_add_ in ascending
point order, then the reference's permute / unbind / cat).  The reference's own op needs its CUDA extension
(bev_pool_ext) and mmcv, both absent: parity unpinned, like spconv and the rotated NMS."""
import numpy as np
import torch


def _torch_bev_pool(x, geom, B, dx, bx, nx):
    """The reference expressions, with index_add_ standing in for sort + segmented sum."""
    Np = x.shape[0]
    x, geom = torch.from_numpy(x), torch.from_numpy(geom)
    dx, bx = torch.tensor(dx, dtype=torch.float32), torch.tensor(bx, dtype=torch.float32)
    g = ((geom - (bx - dx / 2.0)) / dx).long()
    batch_ix = torch.cat([torch.full([Np // B, 1], ix, dtype=torch.long) for ix in range(B)])
    g = torch.cat((g, batch_ix), 1)
    kept = (g[:, 0] >= 0) & (g[:, 0] < nx[0]) & (g[:, 1] >= 0) & (g[:, 1] < nx[1]) & (g[:, 2] >= 0) & (g[:, 2] < nx[2])
    x, g = x[kept], g[kept]
    H, W, Dz, C = nx[0], nx[1], nx[2], x.shape[1]
    out = torch.zeros(B * Dz * H * W, C)
    lin = ((g[:, 3] * Dz + g[:, 2]) * H + g[:, 0]) * W + g[:, 1]           # out[b, z, x, y, c] (bev_pool_cuda.cu:33-36)
    out.index_add_(0, lin, x)
    out = out.view(B, Dz, H, W, C).permute(0, 4, 1, 2, 3).contiguous()     # bev_pool.py:96
    return torch.cat(out.unbind(dim=2), 1)                                  # base.py:161 -> [B, Dz*C, H, W]


def test_oracle_bev_pool_equals_torch_statement(oracle):
    rng = np.random.default_rng(0)
    B, P, C = 2, 6000, 7
    nx, dx, bx = [12, 9, 3], [0.5, 0.75, 2.0], [-2.75, -3.0, -2.0]
    geom = np.stack([rng.uniform(-4.0, 4.0, P), rng.uniform(-4.5, 4.5, P), rng.uniform(-4, 4, P)], 1).astype(np.float32)
    geom[5] = np.nan
    geom[7, 0] = -3.2                       # t in (-1, 0): .long() truncates to cell 0 -- kept by the reference
    x = rng.normal(size=(P, C)).astype(np.float32)
    ref = _torch_bev_pool(x, geom, B, dx, bx, nx).numpy()                   # [B, Dz*C, H, W]
    lo = np.asarray(bx, np.float32) - np.asarray(dx, np.float32) / np.float32(2)
    got = oracle.bev_pool(x, geom, B, lo, dx, nx)                           # [B, H, W, Dz*C]
    assert got.shape == (B, 12, 9, 3 * C)
    assert np.array_equal(got.transpose(0, 3, 1, 2).view(np.int32), ref.view(np.int32))
    assert np.abs(got).sum() > 0


def test_oracle_bev_pool_fused_outer_product_equals_materialised(oracle):
    rng = np.random.default_rng(1)
    B, N, D, fH, fW, C = 2, 3, 5, 4, 6, 8
    depth = rng.uniform(0, 1, (B * N, D, fH, fW)).astype(np.float32)
    ctx = rng.normal(size=(B * N, fH, fW, C)).astype(np.float32)
    x = (depth[..., None] * ctx[:, None]).astype(np.float32)                # depth_lss.py:93: materialised [BN,D,fH,fW,C]
    geom = rng.uniform(-3, 3, (B * N * D * fH * fW, 3)).astype(np.float32)
    lo, dx, nx = [-2, -2, -3], [0.4, 0.4, 6.0], [10, 10, 1]
    a = oracle.bev_pool(x.reshape(-1, C), geom, B, lo, dx, nx)
    b = oracle.bev_pool(ctx.reshape(-1, C), geom, B, lo, dx, nx, depth=depth.reshape(-1), D=D, fHW=fH * fW)
    assert np.array_equal(a.view(np.int32), b.view(np.int32)) and np.abs(a).sum() > 0
