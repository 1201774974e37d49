"""scripts/lab_cold_weights.py for the shapes of the b = 1 fine-tuning step (forward and input-gradient products: N and K swap).

    python scripts/lab_cold_train.py > profiles/r06_lab_cold_train.log
"""
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
sys.argv = [sys.argv[0], "--import-only"]
import scripts.lab_cold_weights as L  # noqa: E402

GV = [(0, 0), (2, 1), (8, 1), (10, 1), (7, 1), (5, 1), (18, 1), (16, 1), (8, 2), (8, 4), (5, 2), (5, 4)]
print("tiles: 0 auto, 2 64x64 register-staged, 8 / 10 64x64 ring of 3 / 5, 7 128x64/3, 5 128x128/2, 18 96x128/2, 16 96x160/2")
SH = [("L0 proj 12288x320x320", 12288, 320, 320), ("L0 qkv 12288x960x320", 12288, 960, 320), ("L0 ff1 12288x2560x320", 12288, 2560, 320),
      ("L0 ff2 12288x320x1280", 12288, 320, 1280), ("L0 dff1 12288x320x2560", 12288, 320, 2560), ("L0 dqkv 12288x320x960", 12288, 320, 960),
      ("L1 proj 3072x640x640", 3072, 640, 640), ("L1 qkv 3072x1920x640", 3072, 1920, 640), ("L1 dqkv 3072x640x1920", 3072, 640, 1920),
      ("L1 ff1 3072x5120x640", 3072, 5120, 640), ("L1 ff2 3072x640x2560", 3072, 640, 2560), ("L1 dff1 3072x640x5120", 3072, 640, 5120),
      ("L1 dff2 3072x2560x640", 3072, 2560, 640),
      ("L2 proj 768x1280x1280", 768, 1280, 1280), ("L2 qkv 768x3840x1280", 768, 3840, 1280), ("L2 dqkv 768x1280x3840", 768, 1280, 3840),
      ("L2 ff1 768x10240x1280", 768, 10240, 1280), ("L2 ff2 768x1280x5120", 768, 1280, 5120), ("L2 dff1 768x1280x10240", 768, 1280, 10240),
      ("L2 dff2 768x5120x1280", 768, 5120, 1280),
      ("mid proj 192x1280x1280", 192, 1280, 1280), ("mid ff1 192x10240x1280", 192, 10240, 1280), ("mid ff2 192x1280x5120", 192, 1280, 5120),
      ("fs qkv 924x2304x768", 924, 2304, 768), ("fs out 924x768x768", 924, 768, 768), ("fs ff1 924x6144x768", 924, 6144, 768),
      ("fs ff2 924x768x3072", 924, 768, 3072), ("fs dff1 924x768x6144", 924, 768, 6144), ("fs dqkv 924x768x2304", 924, 768, 2304)]
for nm, M, N, K in SH:
    L.run("gemm", nm, (M, N, K), GV)
