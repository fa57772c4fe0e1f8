"""Scene description - host-side mirror of the reference's ``utils/mujoco/terrain.py::generate_terrain`` (:309-365).

The reference builds an MJCF scene tree and hands it to MuJoCo; here a scene is a plain dict consumed by the model
marshaller (``cabi.MarshalledModel``).  This round implements the scene the headline benchmark runs on:

* ``flat``: infinite plane named ``floor`` with MuJoCo default contact parameters
  (``assets/scene_flat.xml:31``: ``<geom name="floor" size="0 0 0.05" type="plane"/>``), terrain limits
  ``(10000, -10000, 10000, -10000)`` as ``(max_x, min_x, max_y, min_y)`` (terrain.py:357-359).

``perlin`` / ``random_boxes`` / ``random_pyramids`` and the static ``ramp`` / ``slippery`` / ``stairs`` scenes need
height-field / box narrow-phase kernels and are SURVEY.md §8(f) rank 2 ("next"); asking for them raises.
"""
from __future__ import annotations

_FLOOR_DEFAULT = dict(friction=(1.0, 0.005, 0.0001), margin=0.0, gap=0.0, solmix=1.0, solref=(0.02, 1.0),
                      solimp=(0.9, 0.95, 0.001, 0.5, 2.0), condim=3, priority=0)
_NOT_YET = ('perlin', 'random_boxes', 'random_pyramids', 'ramp', 'slippery', 'stairs')


def generate_terrain(terrain_name: str = 'flat', hip_height: float = 0.3, seed: int = 10):
    """Returns ``(scene_desc, terrain_limits)``; ``seed`` is accepted for signature parity (procedural scenes)."""
    if terrain_name == 'flat':
        return {'name': 'flat', 'floor': dict(_FLOOR_DEFAULT)}, (10000.0, -10000.0, 10000.0, -10000.0)
    if terrain_name in _NOT_YET:
        raise NotImplementedError(f"scene '{terrain_name}' needs the height-field/box narrow phase (SURVEY.md §8f rank 2); "
                                  f"only 'flat' is available on the batched GPU path in this round")
    raise ValueError(f'Invalid scene name: {terrain_name}, available are: flat, random_boxes, random_pyramids, '
                     f'perlin, stairs, ramp, slippery')
