"""Compile the registry robots' MJCFs into the committed data tables gym_quadruped_amd/model_data/*.json.

Run only in the build container (needs the reference checkout for the XML + collision meshes):
    python tools/compile_models.py [/root/reference]
The tables restate the physical parameters of robot_model/<robot>/<robot>.xml;
collision meshes are reduced to their convex-hull vertices.  Nothing else of the reference is read.
"""
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from gym_quadruped_amd.mjcf import compile_mjcf  # noqa: E402

ROBOTS = ['mini_cheetah', 'aliengo', 'go2', 'go1', 'b2', 'hyqreal1', 'hyqreal2', 'spot']


def main():
    ref = Path(sys.argv[1] if len(sys.argv) > 1 else '/root/reference')
    out = Path(__file__).resolve().parents[1] / 'gym_quadruped_amd' / 'model_data'
    out.mkdir(exist_ok=True)
    for r in ROBOTS:
        md = compile_mjcf(ref / 'gym_quadruped' / 'robot_model' / r / f'{r}.xml')
        (out / f'{r}.json').write_text(md.to_json())
        print(f'{r}: nbody={md.nbody} ngeom={md.ngeom} mass={md.total_mass:.3f} nvert={len(md.vert_pos)}')


if __name__ == '__main__':
    main()
