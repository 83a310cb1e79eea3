"""al3d: MI355X-native hot path of the diversity-based active-learning selector.

Scope: the unlabeled-pool sweep (voxelize -> sparse 3-D
encoder -> dense neck -> anchor head/NMS -> BEV embedding) and the
spatial/temporal/feature diversity selectors, behind the reference's registry
API (``build_detector`` / ``build_selector``).  All arithmetic runs in
hand-written HIP kernels behind the C-ABI declared in ``include/al3d.h``;
there is no CPU fallback -- importing ``al3d.lib`` fails loudly if
``libal3d_hip.so`` has not been built.
"""
__version__ = "0.1.0"
