#!/usr/bin/env python3
"""Dev tool: same-box A/B of two builds of libal3d_hip.so.

The one-GPU boxes of the pool differ by ~4 % (power-limited kernels), so two variants are only
comparable inside one gpurun call.  Build both variants locally, keep the two libraries somewhere in
the tree (not under gpurun_out/, which does not travel), then on the box:

    python tools/ab_so.py ab_tmp/lib_a.so ab_tmp/lib_b.so -- python tools/bench_splayers.py 32 al3d_sp_conv_wave2_bf16x6

Each library is copied over csrc/libal3d_hip.so in turn (a, b, a, b) and the command is run as a
child process; the last lines of its output are printed next to the variant name.
"""
import os
import shutil
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    if "--" not in sys.argv or len(sys.argv) < 5:
        sys.exit(__doc__)
    cut = sys.argv.index("--")
    libs, cmd = sys.argv[1:cut], sys.argv[cut + 1:]
    pkg = [d for d in os.listdir(ROOT) if d.endswith("_amd") and os.path.isdir(os.path.join(ROOT, d, "csrc"))][0]
    target = os.path.join(ROOT, pkg, "csrc", "libal3d_hip.so")
    keep = target + ".ab_keep"
    shutil.copy2(target, keep)
    try:
        for rnd in range(2):
            for lib in libs:
                shutil.copy2(lib, target)
                out = subprocess.run(cmd, cwd=ROOT, capture_output=True, text=True)
                tail = [l for l in out.stdout.strip().split("\n") if l][-int(os.environ.get("AB_TAIL", "3")):]
                print(f"== {os.path.basename(lib)} (round {rnd + 1}, rc {out.returncode})")
                for l in tail:
                    print("   " + l[:300])
    finally:
        shutil.move(keep, target)


if __name__ == "__main__":
    main()
