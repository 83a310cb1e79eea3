# Dev tool (CPU, this container): AddressSanitizer + UBSan build of the host JPEG decoder (csrc/jpeg_host.cpp) fed 4,000
# mutations (bit flips, truncations, header bytes overwritten, random runs) of each of four valid files written by Pillow.
#   bash tools/fuzz/run_jpeg_fuzz.sh      -> "fuzz: N decoded, M refused, no crash" (any finding aborts with a report)
set -e
R=$(cd "$(dirname "$0")/../.." && pwd)
SRC=$(ls -d $R/*_amd/csrc)
T=$(mktemp -d)
python3 - "$T" <<'PY'
import sys
import numpy as np
from PIL import Image
rng = np.random.default_rng(0)
yy, xx = np.mgrid[0:97, 0:131]
img = np.stack([(xx * 2) % 256, (yy * 3) % 256, ((xx + yy) * 5) % 256], -1)
img = (img + rng.integers(-20, 20, img.shape)).clip(0, 255).astype(np.uint8)
t = sys.argv[1]
Image.fromarray(img).save(t + "/a420.jpg", quality=85)
Image.fromarray(img).save(t + "/a444.jpg", quality=92, subsampling=0)
Image.fromarray(img).save(t + "/a422.jpg", quality=70, subsampling=1)
Image.fromarray(img[..., 0]).save(t + "/agrey.jpg", quality=80)
PY
g++ -O1 -g -fsanitize=address,undefined -fno-omit-frame-pointer -D__HIP_PLATFORM_AMD__ -I/opt/rocm/include -I$SRC \
    $R/tools/fuzz/jpeg_fuzz.cpp $SRC/jpeg_host.cpp -o $T/fuzz_jpeg
$T/fuzz_jpeg $T/a420.jpg $T/a444.jpg $T/a422.jpg $T/agrey.jpg
rm -rf $T
