"""Scene description - host-side mirror of the reference's ``utils/mujoco/terrain.py::generate_terrain`` (:309-365).

The reference builds an MJCF scene tree and hands it to MuJoCo; here a scene is a plain dict consumed by the model
marshaller (``cabi.MarshalledModel``): the floor plane plus a list of static world boxes.

* ``flat``: infinite plane named ``floor`` with MuJoCo default contact parameters
  (``assets/scene_flat.xml:31``: ``<geom name="floor" size="0 0 0.05" type="plane"/>``), terrain limits
  ``(10000, -10000, 10000, -10000)`` as ``(max_x, min_x, max_y, min_y)`` (terrain.py:357-359).
* ``random_boxes`` (terrain.py:145-238, parameters :324-335) and ``random_pyramids`` (:241-295, :336-344): the same
  draws from numpy's global generator under ``local_seed(seed)`` in the same order, so the box lists are the reference's
  (pinned by ``tests/golden/terrain_boxes.json``, produced by running the reference's own function).
* ``ramp`` / ``slippery`` / ``stairs``: the box geoms of the reference's static scene files, shipped as data
  (``model_data/static_scenes.json``, extracted by ``tools/gen_golden_terrain.py``).
* ``perlin`` (terrain.py:26-118, parameters :345-356): a 128 x 128 height field of ``noise.pnoise2`` samples over the flat
  floor.  ``noise`` (Casey Duncan's package, C extension ``_perlin``) is third-party and absent here; ``pnoise2`` below
  restates its published algorithm (Ken Perlin's improved noise with the classic permutation table, fp32 arithmetic, octave
  sum normalised by the amplitude sum) and is pinned by the image the reference ships
  (``robot_model/mini_cheetah/height_field.png``, reproduced bit for bit with add_perlin_heightfield's default arguments:
  ``tests/golden/perlin_default.npz``).  The reference writes the image to a PNG and MuJoCo's compiler loads it: rows
  flipped, elevation shifted / scaled to [0, 1] (from MuJoCo's documentation, unverified here - no MuJoCo).

Scene generation is host-side set-up code; the boxes and the height field travel in ``GqModelDesc`` (``box_*`` tables,
``hfield_*``) and are simulated by the BOXES variants of the step kernel (``csrc/gq_boxes.h``; Newton solver only -
``gq_model_create`` rejects them with PGS).
"""
from __future__ import annotations

import contextlib
import json
from pathlib import Path

import numpy as np
from scipy.spatial.transform import Rotation

_FLOOR_DEFAULT = dict(friction=(1.0, 0.005, 0.0001), margin=0.0, gap=0.0, solmix=1.0, solref=(0.02, 1.0),
                      solimp=(0.9, 0.95, 0.001, 0.5, 2.0), condim=3, priority=0)
_BOX_DEFAULT = dict(friction=(1.0, 0.005, 0.0001), margin=0.0, gap=0.0, solmix=1.0, solref=(0.02, 1.0),
                    solimp=(0.9, 0.95, 0.001, 0.5, 2.0), condim=3, priority=0)
_STATIC = ('ramp', 'slippery', 'stairs')
_FLAT_LIMITS = (10000.0, -10000.0, 10000.0, -10000.0)


_PERM = np.array([
    151, 160, 137, 91, 90, 15, 131, 13, 201, 95, 96, 53, 194, 233, 7, 225, 140, 36, 103, 30, 69, 142, 8, 99, 37, 240, 21, 10, 23, 190, 6,
    148, 247, 120, 234, 75, 0, 26, 197, 62, 94, 252, 219, 203, 117, 35, 11, 32, 57, 177, 33, 88, 237, 149, 56, 87, 174, 20, 125, 136,
    171, 168, 68, 175, 74, 165, 71, 134, 139, 48, 27, 166, 77, 146, 158, 231, 83, 111, 229, 122, 60, 211, 133, 230, 220, 105, 92, 41, 55,
    46, 245, 40, 244, 102, 143, 54, 65, 25, 63, 161, 1, 216, 80, 73, 209, 76, 132, 187, 208, 89, 18, 169, 200, 196, 135, 130, 116, 188,
    159, 86, 164, 100, 109, 198, 173, 186, 3, 64, 52, 217, 226, 250, 124, 123, 5, 202, 38, 147, 118, 126, 255, 82, 85, 212, 207, 206, 59,
    227, 47, 16, 58, 17, 182, 189, 28, 42, 223, 183, 170, 213, 119, 248, 152, 2, 44, 154, 163, 70, 221, 153, 101, 155, 167, 43, 172, 9,
    129, 22, 39, 253, 19, 98, 108, 110, 79, 113, 224, 232, 178, 185, 112, 104, 218, 246, 97, 228, 251, 34, 242, 193, 238, 210, 144, 12,
    191, 179, 162, 241, 81, 51, 145, 235, 249, 14, 239, 107, 49, 192, 214, 31, 181, 199, 106, 157, 184, 84, 204, 176, 115, 121, 50, 45,
    127, 4, 150, 254, 138, 236, 205, 93, 222, 114, 67, 29, 24, 72, 243, 141, 128, 195, 78, 66, 215, 61, 156, 180] * 2, dtype=np.int64)
_GRAD_XY = np.array([[1, 1], [-1, 1], [1, -1], [-1, -1], [1, 0], [-1, 0], [1, 0], [-1, 0], [0, 1], [0, -1], [0, 1], [0, -1],
                     [1, 0], [-1, 0], [0, -1], [0, 1]], dtype=np.float32)   # x, y of the 16 gradient directions


def _noise2(x, y, repeatx, repeaty, base=0):
    """One octave of 2-D improved Perlin noise on float32 arrays (the `noise` package's noise2)."""
    f32 = np.float32
    i = np.floor(np.fmod(x, f32(repeatx))).astype(np.int64)
    j = np.floor(np.fmod(y, f32(repeaty))).astype(np.int64)
    ii = np.fmod((i + 1).astype(f32), f32(repeatx)).astype(np.int64)
    jj = np.fmod((j + 1).astype(f32), f32(repeaty)).astype(np.int64)
    i, j, ii, jj = (i & 255) + base, (j & 255) + base, (ii & 255) + base, (jj & 255) + base
    x = x - np.floor(x)
    y = y - np.floor(y)
    fx = x * x * x * (x * (x * f32(6) - f32(15)) + f32(10))
    fy = y * y * y * (y * (y * f32(6) - f32(15)) + f32(10))
    a, b = _PERM[i], _PERM[ii]
    aa, ab, ba, bb = _PERM[a + j], _PERM[a + jj], _PERM[b + j], _PERM[b + jj]

    def grad(h, xx, yy):
        g = _GRAD_XY[h & 15]
        return xx * g[..., 0] + yy * g[..., 1]

    def lerp(t, p, q):
        return p + t * (q - p)

    return lerp(fy, lerp(fx, grad(_PERM[aa], x, y), grad(_PERM[ba], x - 1, y)),
                lerp(fx, grad(_PERM[ab], x, y - 1), grad(_PERM[bb], x - 1, y - 1)))


def pnoise2(x, y, octaves=1, persistence=0.5, lacunarity=2.0, repeatx=1024.0, repeaty=1024.0, base=0):
    """``noise.pnoise2`` on arrays: octaves of :func:`_noise2`, frequency x lacunarity and amplitude x persistence per
    octave, the sum divided by the sum of the amplitudes; float32 throughout, like the C extension."""
    f32 = np.float32
    x, y = np.asarray(x, dtype=f32), np.asarray(y, dtype=f32)
    if octaves == 1:
        return _noise2(x, y, repeatx, repeaty, base)
    freq, amp, total_amp, total = f32(1), f32(1), f32(0), np.zeros(np.broadcast(x, y).shape, dtype=f32)
    for _ in range(int(octaves)):
        total = total + _noise2(x * freq, y * freq, f32(repeatx) * freq, f32(repeaty) * freq, base) * amp
        total_amp = f32(total_amp + amp)
        freq = f32(freq * f32(lacunarity))
        amp = f32(amp * f32(persistence))
    return total / total_amp


def perlin_image(image_width=128, img_height=128, smooth=100.0, perlin_octaves=6, perlin_persistence=0.5, perlin_lacunarity=2.0):
    """The uint8 image add_perlin_heightfield writes (terrain.py:75-86): ``image[y, x] = int((pnoise2(x / smooth,
    y / smooth, ...) + 1) / 2 * 255)``; the reference loops ``range(image_width)`` for both axes."""
    xs, ys = np.meshgrid(np.arange(image_width, dtype=np.float64), np.arange(image_width, dtype=np.float64))
    nv = pnoise2(xs / smooth, ys / smooth, perlin_octaves, perlin_persistence, perlin_lacunarity).astype(np.float64)
    img = np.zeros((img_height, image_width), dtype=np.uint8)
    img[:image_width, :] = ((nv + 1.0) / 2.0 * 255.0).astype(np.int64).astype(np.uint8)[:img_height]
    return img


def _perlin_heightfield(size, max_height, min_height, position=(0.0, 0.0, 0.0), **image_kwargs):
    """Height-field description + terrain limits of add_perlin_heightfield (terrain.py:26-118): hfield ``size`` =
    (size_x / 2, size_y / 2, max_height, min_height); MuJoCo's compiler flips the image rows (row 0 of the data is the
    image's last row, at y = -size_y / 2) and maps the elevation to [0, 1]."""
    img = perlin_image(**image_kwargs).astype(np.float64)[::-1]
    lo, hi = img.min(), img.max()
    data = (img - lo) / (hi - lo) if hi > lo else np.zeros_like(img)
    hfield = dict(_BOX_DEFAULT)
    hfield.update(data=data.astype(np.float32), size=(size[0] / 2.0, size[1] / 2.0, float(max_height), float(min_height)),
                  pos=tuple(float(v) for v in position))
    cx, cy = position[0], position[1]
    mx, my = size[0] / 2.0, size[1] / 2.0
    radius = 0.8 * np.sqrt((mx - cx) * (mx - cx)) if mx >= my else 0.8 * np.sqrt((my - cy) * (my - cy))
    return hfield, (cx + radius, cx - radius, cy + radius, cy - radius)


@contextlib.contextmanager
def local_seed(seed):
    """numpy's GLOBAL generator seeded for the duration of the block, restored afterwards (terrain.py:299-306)."""
    state = np.random.get_state()
    np.random.seed(seed)
    try:
        yield
    finally:
        np.random.set_state(state)


def _box(pos, euler, size, **over):
    """World box from a centre, xyz Euler angles and FULL extents (add_box, terrain.py:121-142: MuJoCo takes half sizes)."""
    quat = Rotation.from_euler('xyz', euler).as_quat(canonical=True, scalar_first=True)
    b = dict(_BOX_DEFAULT)
    b.update(pos=[float(x) for x in pos], size=[0.5 * float(x) for x in size], quat=[float(x) for x in quat])
    b.update(over)
    return b


def _world_of_boxes(init_pos, euler, nums, box_size, box_euler, separation, box_size_rand, box_euler_rand, separation_rand,
                    random_roll_pitch):
    """Grid of randomly sized / oriented boxes (add_world_of_boxes, terrain.py:145-238).  Draw order per box: size xy (2),
    size z (1), Euler angles (3, or yaw only), x separation (1), y separation (1); two draws for the first column step."""
    U = np.random.uniform
    init_pos, separation, separation_rand = np.asarray(init_pos, float), np.asarray(separation, float), np.asarray(separation_rand, float)
    box_size, box_size_rand, box_euler_rand = np.asarray(box_size, float), np.asarray(box_size_rand, float), np.asarray(box_euler_rand, float)
    Rw = Rotation.from_euler('xyz', euler).as_matrix()
    boxes, local = [], np.zeros(3)
    ext = [0.0, 0, 0.0, 0]   # largest |x|, its sign, largest |y|, its sign
    step = separation + separation_rand * U(-1.0, 1.0, 2)
    for _ in range(nums[0]):
        local[0] += step[0]
        local[1] = 0.0
        for _ in range(nums[1]):
            size_xy = box_size[0:2] + box_size_rand[0:2] * U(-0.2, 0.2, 2)
            size_z = box_size[2] + box_size_rand[2] * U(-0.1, 0.15, 1)
            if random_roll_pitch:
                ang = np.asarray(box_euler, float) + box_euler_rand * U(-1.0, 1.0, 3)
            else:
                ang = np.array(box_euler, float)
                ang[2] = ang[2] + box_euler_rand[2] * U(-1, 1, 1)[0]
            step = np.array([separation[0] + separation_rand[0] * U(0, 0.5, 1)[0], separation[1] + separation_rand[1] * U(-0.5, 0.5, 1)[0]])
            local[1] += step[1]
            boxes.append(_box(Rw @ local + init_pos, ang, [size_xy[0], size_xy[1], size_z[0]]))
            ax, ay = abs(local[0] + init_pos[0]), abs(local[1] + init_pos[1])
            if ax >= ext[0]:
                ext[0], ext[1] = ax, (1 if ax > 0 else -1)
            if ay >= ext[2]:
                ext[2], ext[3] = ay, (1 if ay > 0 else -1)
    max_x, max_y = ext[0] * ext[1], ext[2] * ext[3]
    cx, cy = (max_x + init_pos[0]) / 2, (max_y + init_pos[1]) / 2
    radius = 1.2 * np.sqrt(2 * (max_x - cx) ** 2) if ext[0] >= ext[2] else 1.2 * np.sqrt(2 * (max_y - cy) ** 2)
    return boxes, (float(cx + radius), float(cx - radius), float(cy + radius), float(cy - radius))


def _world_of_pyramid(init_pos, yaw, width, max_height, length, stair_nums):
    """Stack of shrinking slabs (add_world_of_pyramid, terrain.py:241-295); two draws: slab height, stride."""
    U = np.random.uniform
    boxes, local = [], np.array([0.0, 0.0, -0.05])
    height = U(0.08, max_height, 1)[0]
    stride = U(0.5, 1.0, 1)[0]
    Rz = Rotation.from_euler('xyz', [0, 0, yaw]).as_matrix()
    mx = my = 0.0
    center = (0.0, 0.0)
    for i in range(int(stair_nums)):
        local[2] += height
        x, y, _ = Rz @ local
        w, l = width - stride * i, length - stride * i
        if w < 0.3 or l < 0.3:
            break
        boxes.append(_box([x + init_pos[0], y + init_pos[1], local[2]], [0.0, 0.0, yaw], [w, l, height]))
        if i == 0:
            mx, my = abs(x + init_pos[0] + w / 2.0), abs(y + init_pos[1] + l / 2.0)
            center = (x + init_pos[0], y + init_pos[1])
    radius = 1.5 * np.sqrt(2 * (mx - center[0]) ** 2) if mx >= my else 1.5 * np.sqrt(2 * (my - center[1]) ** 2)
    return boxes, (float(center[0] + radius), float(center[0] - radius), float(center[1] + radius), float(center[1] - radius))


def generate_terrain(terrain_name: str = 'flat', hip_height: float = 0.3, seed: int = 10):
    """Returns ``(scene_desc, terrain_limits)``: ``scene_desc = {'name', 'floor': {...}, 'boxes': [{pos, size (half
    extents), quat (wxyz), friction, priority, condim, solref, solimp, solmix, margin, gap}, ...], 'hfield': {data [nrow][ncol]
    in [0, 1], size (rx, ry, elevation, base), pos, + contact parameters} (perlin only)}``."""
    scene = {'name': terrain_name, 'floor': dict(_FLOOR_DEFAULT), 'boxes': []}
    if terrain_name == 'flat':
        return scene, _FLAT_LIMITS
    if terrain_name in _STATIC:   # robot_model/scene_<name>.xml exists in the reference: static file, flat limits (:319-321)
        data = json.loads((Path(__file__).parent / 'model_data' / 'static_scenes.json').read_text())[terrain_name]
        for b in data['boxes']:
            box = dict(_BOX_DEFAULT)
            box.update({k: v for k, v in b.items() if k != 'name'})
            box['friction'] = tuple(b.get('friction', _BOX_DEFAULT['friction']))
            scene['boxes'].append(box)
        return scene, _FLAT_LIMITS
    with local_seed(seed):
        h = float(hip_height)
        if terrain_name == 'random_boxes':      # parameters of terrain.py:324-335
            scene['boxes'], limits = _world_of_boxes(
                init_pos=[0.5, -3.0, 0.02], euler=[0, 0, 0.0], nums=[10, 10], box_size=[2 * h, 2 * h, h / 2.0], box_euler=[0.0, 0.0, 0.0],
                separation=[2 * h, 2 * h], box_size_rand=[0.5 * h, 0.5 * h, h / 2], box_euler_rand=[0.1, 0.1, 2 * np.pi],
                separation_rand=[0, 1], random_roll_pitch=True)
            return scene, limits
        if terrain_name == 'random_pyramids':   # :336-344; the stair count is drawn before the slab height and stride
            stair_nums = np.random.uniform(2, 8, 1)[0]
            scene['boxes'], limits = _world_of_pyramid(init_pos=[3, 0, 0.02], yaw=0.0, width=10 * h, max_height=5 * h, length=10 * h,
                                                       stair_nums=stair_nums)
            return scene, limits
    if terrain_name == 'perlin':            # :345-356
        h = float(hip_height)
        scene['hfield'], limits = _perlin_heightfield(size=(h * 100, h * 100), max_height=2 * h, min_height=0.005, image_width=128,
                                                      img_height=128, smooth=50, perlin_octaves=5, perlin_lacunarity=4.0)
        return scene, tuple(float(v) for v in limits)
    raise ValueError(f'Invalid scene name: {terrain_name}, available are: flat, random_boxes, random_pyramids, '
                     f'perlin, stairs, ramp, slippery')
