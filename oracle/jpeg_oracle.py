"""TEST INFRASTRUCTURE ONLY -- never imported by the product path.

numpy restatement of what follows the entropy decoder in libjpeg-turbo at its defaults (the decoder Pillow's ``Image.open``
runs in the reference's loader, bevfusion/mmdet3d/datasets/pipelines/loading.py:19-58; libjpeg-turbo is Pillow's dependency,
absent from /root/reference: restated from its published algorithms, pinned against the INSTALLED Pillow's bytes in
tests/test_jpeg_host.py):

  * ``idct_islow``      jidctint.c jpeg_idct_islow (13-bit fixed point, PASS1_BITS 2), +128, clamp
  * ``upsample``        jdsample.c fancy upsampling h2v1 / h2v2 / h1v2 with the library's edge rules
  * ``ycc_to_rgb``      jdcolor.c 16-bit fixed-point tables

``decode(info, quant, coefs)`` turns the C entropy decoder's output (al3d_jpeg_entropy_decode) into [H, W, 3] uint8.
"""
import numpy as np

F = dict(c0298=2446, c0390=3196, c0541=4433, c0765=6270, c0899=7373, c1175=9633, c1501=12299, c1847=15137, c1961=16069,
         c2053=16819, c2562=20995, c3072=25172)


def _descale(x, n):
    return (x + (1 << (n - 1))) >> n


def _idct8(v, shift):
    """v [..., 8] int64 -> one 1-D pass of jpeg_idct_islow along the last axis."""
    z2, z3 = v[..., 2], v[..., 6]
    z1 = (z2 + z3) * F["c0541"]
    tmp2 = z1 + z3 * (-F["c1847"])
    tmp3 = z1 + z2 * F["c0765"]
    z2, z3 = v[..., 0], v[..., 4]
    tmp0, tmp1 = (z2 + z3) << 13, (z2 - z3) << 13
    tmp10, tmp13, tmp11, tmp12 = tmp0 + tmp3, tmp0 - tmp3, tmp1 + tmp2, tmp1 - tmp2
    t0, t1, t2, t3 = v[..., 7], v[..., 5], v[..., 3], v[..., 1]
    z1, z2, z3, z4 = t0 + t3, t1 + t2, t0 + t2, t1 + t3
    z5 = (z3 + z4) * F["c1175"]
    t0, t1, t2, t3 = t0 * F["c0298"], t1 * F["c2053"], t2 * F["c3072"], t3 * F["c1501"]
    z1, z2, z3, z4 = z1 * -F["c0899"], z2 * -F["c2562"], z3 * -F["c1961"] + z5, z4 * -F["c0390"] + z5
    t0, t1, t2, t3 = t0 + z1 + z3, t1 + z2 + z4, t2 + z2 + z3, t3 + z1 + z4
    out = np.stack([tmp10 + t3, tmp11 + t2, tmp12 + t1, tmp13 + t0, tmp13 - t0, tmp12 - t1, tmp11 - t2, tmp10 - t3], -1)
    return _descale(out, shift)


def idct_islow(coefs, quant):
    """coefs [nblocks, 64] int16 (natural order), quant [64] -> samples [nblocks, 8, 8] uint8."""
    x = coefs.astype(np.int64).reshape(-1, 8, 8) * quant.astype(np.int64).reshape(8, 8)
    ws = np.swapaxes(_idct8(np.swapaxes(x, 1, 2), 13 - 2), 1, 2)          # pass 1: columns
    out = _idct8(ws, 13 + 2 + 3)                                          # pass 2: rows
    return np.clip(out + 128, 0, 255).astype(np.uint8)


def plane(blocks, bh, bw):
    return blocks.reshape(bh, bw, 8, 8).transpose(0, 2, 1, 3).reshape(bh * 8, bw * 8)


def upsample(pl, dw, dh, hr, vr, W, H):
    """pl: the component's sample plane (>= dh x dw real samples) -> [H, W] int64 at full resolution."""
    p = pl[:dh, :dw].astype(np.int64)
    if hr == 1 and vr == 1:
        return p[:H, :W]
    if vr == 2:                                   # rows: nearer / farther with replicas beyond the first / last real row
        up = np.concatenate([p[:1], p[:-1]], 0)
        dn = np.concatenate([p[1:], p[-1:]], 0)
        if hr == 1:                               # h1v2
            rows = np.empty((2 * dh, dw), np.int64)
            rows[0::2] = (3 * p + up + 1) >> 2
            rows[1::2] = (3 * p + dn + 2) >> 2
            return rows[:H, :W]
        near = np.repeat(p, 2, 0)
        far = np.empty((2 * dh, dw), np.int64)
        far[0::2], far[1::2] = up, dn
        s = 3 * near + far                        # column sums of h2v2
        out = np.empty((2 * dh, 2 * dw), np.int64)
        last = np.concatenate([s[:, :1], s[:, :-1]], 1)
        nxt = np.concatenate([s[:, 1:], s[:, -1:]], 1)
        out[:, 0::2] = (3 * s + last + 8) >> 4
        out[:, 1::2] = (3 * s + nxt + 7) >> 4
        out[:, 0] = (4 * s[:, 0] + 8) >> 4
        out[:, 2 * dw - 1] = (4 * s[:, -1] + 7) >> 4
        return out[:H, :W]
    out = np.empty((dh, 2 * dw), np.int64)        # h2v1
    last = np.concatenate([p[:, :1], p[:, :-1]], 1)
    nxt = np.concatenate([p[:, 1:], p[:, -1:]], 1)
    out[:, 0::2] = (3 * p + last + 1) >> 2
    out[:, 1::2] = (3 * p + nxt + 2) >> 2
    out[:, 0] = p[:, 0]
    out[:, 2 * dw - 1] = p[:, -1]
    return out[:H, :W]


def ycc_to_rgb(y, cb, cr):
    cb, cr = cb - 128, cr - 128
    r = y + ((91881 * cr + 32768) >> 16)
    b = y + ((116130 * cb + 32768) >> 16)
    g = y + ((-22554 * cb + 32768 - 46802 * cr) >> 16)
    return np.clip(np.stack([r, g, b], -1), 0, 255).astype(np.uint8)


def decode(info, quant, coefs):
    W, H, nc = int(info[0]), int(info[1]), int(info[2])
    mh, mv = int(info[22]), int(info[23])
    comps = []
    for c in range(nc):
        h, v, bw, bh, off = int(info[3 + c]), int(info[6 + c]), int(info[11 + c]), int(info[14 + c]), int(info[17 + c])
        pl = plane(idct_islow(coefs[off:off + bw * bh], quant[c]), bh, bw)
        dw, dh = -(-W * h // mh), -(-H * v // mv)
        comps.append(upsample(pl, dw, dh, mh // h, mv // v, W, H))
    if nc == 1:
        return np.repeat(comps[0].astype(np.uint8)[..., None], 3, -1)
    return ycc_to_rgb(*comps)
