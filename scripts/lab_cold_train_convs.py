"""scripts/lab_cold_weights.py for the 3x3 convs of the b = 1 fine-tuning step / of one CFG half (12 frames: 12 288 / 3 072 / 768 / 192 rows;
forward and input-gradient convs have the same shapes with Ci and Co swapped).

    python scripts/lab_cold_train_convs.py > profiles/r06_lab_cold_train_convs.log
"""
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
sys.argv = [sys.argv[0], "--import-only"]
import scripts.lab_cold_weights as L  # noqa: E402

print("tiles: 0 auto, 5 128x128/2, 16 96x160/2, 18 96x128/2, 8 64x64/3")
V0 = [(0, 0), (16, 1), (16, 2), (5, 1), (5, 2), (18, 1), (18, 2)]
for nm, d in (("conv 32x32 320->320", (12, 32, 320, 320)), ("conv 32x32 640->320", (12, 32, 640, 320)), ("conv 32x32 320->640", (12, 32, 320, 640)),
              ("conv 32x32 960->320", (12, 32, 960, 320))):
    L.run("conv", nm, d, V0)
V1 = [(0, 0), (5, 2), (5, 4), (5, 8), (16, 2), (16, 4), (16, 8), (18, 2), (18, 4), (8, 4), (8, 8)]
for nm, d in (("conv 16x16 640->640", (12, 16, 640, 640)), ("conv 16x16 1280->640", (12, 16, 1280, 640)), ("conv 16x16 640->1280", (12, 16, 640, 1280)),
              ("conv 16x16 320->640", (12, 16, 320, 640)), ("conv 16x16 640->320", (12, 16, 640, 320)), ("conv 16x16 1920->640", (12, 16, 1920, 640))):
    L.run("conv", nm, d, V1)
V2 = [(0, 0), (5, 2), (5, 4), (5, 8), (5, 16), (16, 2), (16, 4), (16, 8), (18, 4), (8, 8), (8, 16)]
for nm, d in (("conv 8x8 1280->1280", (12, 8, 1280, 1280)), ("conv 8x8 2560->1280", (12, 8, 2560, 1280)), ("conv 8x8 1280->2560", (12, 8, 1280, 2560)),
              ("conv 8x8 640->1280", (12, 8, 640, 1280)), ("conv 8x8 1280->640", (12, 8, 1280, 640))):
    L.run("conv", nm, d, V2)
V3 = [(0, 0), (5, 8), (5, 16), (16, 8), (16, 16), (8, 8), (8, 16), (8, 32)]
for nm, d in (("conv 4x4 1280->1280", (12, 4, 1280, 1280)), ("conv 4x4 2560->1280", (12, 4, 2560, 1280)), ("conv 4x4 1280->2560", (12, 4, 1280, 2560))):
    L.run("conv", nm, d, V3)
