"""Box coder objects configs pass to the head (reference det3d/core/bbox/box_coders.py:40-109,
det3d/builder.py:399-432).  Decoding itself happens inside the head's device kernel; the
object carries the settings the reference's ``MultiGroupHead`` reads (``code_size``, ``n_dim``)."""


class GroundBox3dCoderTorch:
    def __init__(self, linear_dim=False, vec_encode=False, n_dim=7, norm_velo=False):
        self.linear_dim = linear_dim
        self.vec_encode = vec_encode
        self.norm_velo = norm_velo
        self.n_dim = n_dim

    @property
    def code_size(self):
        return self.n_dim + 1 if self.vec_encode else self.n_dim


def build_box_coder(cfg):
    if cfg["type"] != "ground_box3d_coder":
        raise ValueError("unknown box_coder type")
    return GroundBox3dCoderTorch(cfg["linear_dim"], cfg["encode_angle_vector"],
                                 n_dim=cfg.get("n_dim", 9), norm_velo=cfg.get("norm_velo", False))
