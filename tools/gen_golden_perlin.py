"""Golden vector for the perlin scene: the 128 x 128 uint8 image the reference ships as
robot_model/mini_cheetah/height_field.png - an output of its add_perlin_heightfield (terrain.py:26-118) with that function's
default arguments (smooth 100, 6 octaves, persistence 0.5, lacunarity 2.0), i.e. of the third-party noise.pnoise2 that is
not installed here.  Runs in the build container only (reads /root/reference); writes tests/golden/perlin_default.npz."""
from pathlib import Path
import numpy as np
from PIL import Image

ROOT = Path(__file__).resolve().parents[1]
img = np.array(Image.open('/root/reference/gym_quadruped/robot_model/mini_cheetah/height_field.png'))
assert img.shape == (128, 128) and img.dtype == np.uint8
np.savez_compressed(ROOT / 'tests' / 'golden' / 'perlin_default.npz', image=img, smooth=100.0, octaves=6, persistence=0.5, lacunarity=2.0)
print('wrote perlin_default.npz', img.min(), img.max())
