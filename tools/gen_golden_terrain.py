"""Golden vectors for the procedural / static box scenes: the reference's OWN `generate_terrain` run in this container
(under the import stubs of tools/gen_golden.py), its box list and terrain limits dumped as data.

    python tools/gen_golden_terrain.py        # -> tests/golden/terrain_boxes.json, gym_quadruped_amd/model_data/static_scenes.json

tests/golden/terrain_boxes.json: for (scene, hip_height) pairs the list of world box geoms (pos, size = half extents,
quat, friction / priority when given) and terrain_limits the reference produces with seed 10.  static_scenes.json: the
box geoms of robot_model/scene_{ramp,slippery,stairs}.xml (scene data the batched env loads instead of the XML).
/root/reference never travels; these files do."""
import json, sys, types
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[1]
REF = Path('/root/reference')
sys.path.insert(0, str(REF))


class _Any:
    def __getattr__(self, k): return self
    def __call__(self, *a, **k): return self


for name in ('cv2', 'noise', 'mujoco', 'mujoco.viewer', 'gymnasium', 'gymnasium.spaces'):
    m = types.ModuleType(name)
    m.__getattr__ = lambda k, _a=_Any(): _a
    sys.modules[name] = m

from gym_quadruped.utils.mujoco import terrain  # noqa: E402


def boxes_of(tree):
    out = []
    for g in tree.getroot().find('worldbody').iter('geom'):
        if g.get('type') != 'box':
            continue
        f = lambda k, d: [float(x) for x in g.get(k, d).split()]
        b = dict(pos=f('pos', '0 0 0'), size=f('size', '0 0 0'), quat=f('quat', '1 0 0 0'))
        for k in ('friction',):
            if g.get(k): b[k] = f(k, '')
        for k in ('priority', 'condim'):
            if g.get(k): b[k] = int(g.get(k))
        if g.get('name'): b['name'] = g.get('name')
        out.append(b)
    return out


assets = REF / 'gym_quadruped' / 'utils' / 'mujoco' / 'assets'
golden = {}
for scene in ('random_boxes', 'random_pyramids'):
    for hip in (0.225, 0.3, 0.5):
        tree, lim = terrain.generate_terrain(REF / 'gym_quadruped' / 'robot_model' / f'scene_{scene}.xml', assets, hip, scene, seed=10)
        golden[f'{scene}@{hip}'] = dict(boxes=boxes_of(tree), terrain_limits=[float(x) for x in lim])
(ROOT / 'tests' / 'golden' / 'terrain_boxes.json').write_text(json.dumps(golden))
static = {}
for scene in ('ramp', 'slippery', 'stairs'):
    tree, lim = terrain.generate_terrain(REF / 'gym_quadruped' / 'robot_model' / f'scene_{scene}.xml', assets, 0.3, scene, seed=10)
    static[scene] = dict(boxes=boxes_of(tree), terrain_limits=[float(x) for x in lim])
(ROOT / 'gym_quadruped_amd' / 'model_data' / 'static_scenes.json').write_text(json.dumps(static, indent=1))
print({k: (len(v['boxes']), [round(x, 3) for x in v['terrain_limits']]) for k, v in {**golden, **static}.items()})
