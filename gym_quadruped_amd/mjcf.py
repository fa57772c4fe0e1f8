"""MJCF-subset model compiler (host side, numpy, fp64).

The reference builds its physics model with ``mujoco.MjModel.from_xml_path``
(reference ``gym_quadruped/quadruped_env.py:170``).  MuJoCo is a third-party
dependency that is not part of the reference tree, so this module restates the
part of MuJoCo's model compiler that the eight quadruped MJCFs exercise
(SURVEY.md §7 step 1, Appendix B): nested ``<default class>`` inheritance and
``childclass``, the body tree, explicit ``<inertial>``, free + hinge joints,
collision geoms (plane / sphere / capsule / cylinder / box / mesh), sites,
torque motors, a few sensors, keyframes and ``<include>``.

The output is a :class:`ModelDesc` - flat numpy arrays in MuJoCo's own naming
(``body_parentid``, ``jnt_axis``, ``dof_invweight0`` ...) - which is what both
the C oracle (``oracle/``) and the HIP library (``csrc/``) consume through
their C-ABI ``GqModelDesc``.  Compiled descriptions of the registry robots are
committed as data tables under ``model_data/`` so that nothing under
``/root/reference`` is needed at run time.

Collision geometry attached to the robot is lowered to *vertex clouds with a
radius* (sphere = 1 vertex + r, capsule = 2 + r, box = 8 corners, mesh = convex
hull vertices, cylinder = two 16-gon rims).  Against a plane (and later a
height field) the signed distance of such a geom is ``min_v n.(v - p0) - r``,
which is exact for sphere/capsule/box/convex mesh.
"""
from __future__ import annotations

import dataclasses
import json
import os
import xml.etree.ElementTree as ET
from pathlib import Path

import numpy as np

MJ_MINVAL = 1e-15

GEOM_PLANE, GEOM_HFIELD, GEOM_SPHERE, GEOM_CAPSULE, GEOM_ELLIPSOID, GEOM_CYLINDER, GEOM_BOX, GEOM_MESH = range(8)
_GEOM_TYPES = {
    'plane': GEOM_PLANE, 'hfield': GEOM_HFIELD, 'sphere': GEOM_SPHERE, 'capsule': GEOM_CAPSULE,
    'ellipsoid': GEOM_ELLIPSOID, 'cylinder': GEOM_CYLINDER, 'box': GEOM_BOX, 'mesh': GEOM_MESH,
}
JNT_FREE, JNT_BALL, JNT_SLIDE, JNT_HINGE = range(4)

# MuJoCo element defaults (MuJoCo XML reference; Appendix A of SURVEY.md)
_GEOM_DEF = dict(type='sphere', size='0 0 0', pos='0 0 0', quat='1 0 0 0', friction='1 0.005 0.0001',
                 margin='0', gap='0', condim='3', contype='1', conaffinity='1', priority='0',
                 solref='0.02 1', solimp='0.9 0.95 0.001 0.5 2', solmix='1', group='0', density='1000')
_JOINT_DEF = dict(type='hinge', pos='0 0 0', axis='0 0 1', range='0 0', limited='auto', damping='0',
                  armature='0', frictionloss='0', stiffness='0', margin='0', ref='0',
                  solreflimit='0.02 1', solimplimit='0.9 0.95 0.001 0.5 2',
                  solreffriction='0.02 1', solimpfriction='0.9 0.95 0.001 0.5 2',
                  actuatorfrcrange='0 0', actuatorfrclimited='auto')
_MOTOR_DEF = dict(ctrlrange='0 0', ctrllimited='auto', forcerange='0 0', forcelimited='auto', gear='1 0 0 0 0 0')
_SITE_DEF = dict(pos='0 0 0', quat='1 0 0 0')


def _f(s, n=None):
    a = np.array([float(x) for x in str(s).split()], dtype=np.float64)
    if n is not None and a.size < n:
        a = np.concatenate([a, np.zeros(n - a.size)])
    return a


# ----------------------------------------------------------------------------- quaternion helpers (wxyz)
def quat_mul(a, b):
    return np.array([
        a[0] * b[0] - a[1] * b[1] - a[2] * b[2] - a[3] * b[3],
        a[0] * b[1] + a[1] * b[0] + a[2] * b[3] - a[3] * b[2],
        a[0] * b[2] - a[1] * b[3] + a[2] * b[0] + a[3] * b[1],
        a[0] * b[3] + a[1] * b[2] - a[2] * b[1] + a[3] * b[0],
    ])


def quat_to_mat(q):
    w, x, y, z = q / np.linalg.norm(q)
    return np.array([
        [1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y)],
        [2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x)],
        [2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)],
    ])


def quat_z_to_vec(v):
    """Quaternion rotating the z axis onto unit vector v (MuJoCo mju_quatZ2Vec)."""
    v = v / np.linalg.norm(v)
    z = np.array([0.0, 0.0, 1.0])
    axis = np.cross(z, v)
    s = np.linalg.norm(axis)
    if s < 1e-10:
        return np.array([1.0, 0, 0, 0]) if v[2] > 0 else np.array([0.0, 1.0, 0, 0])
    axis /= s
    ang = np.arctan2(s, v[2])
    return np.concatenate([[np.cos(ang / 2)], np.sin(ang / 2) * axis])


def axis_angle_quat(axis, angle):
    return np.concatenate([[np.cos(angle / 2)], np.sin(angle / 2) * np.asarray(axis)])


# ----------------------------------------------------------------------------- defaults
class _Defaults:
    """Nested <default class=...> tree: class name -> {tag: attrib dict} with parent inheritance."""

    def __init__(self):
        self.classes: dict[str, dict[str, dict[str, str]]] = {'main': {}}

    def load(self, elem, parent='main', top=True):
        name = elem.attrib.get('class', 'main' if top else None)
        if name is None:
            raise ValueError('nested <default> needs a class name')
        if name not in self.classes:
            self.classes[name] = {k: dict(v) for k, v in self.classes[parent].items()}
        cur = self.classes[name]
        for ch in elem:
            if ch.tag == 'default':
                continue
            cur.setdefault(ch.tag, {}).update(ch.attrib)
        for ch in elem:
            if ch.tag == 'default':
                self.load(ch, parent=name, top=False)

    def resolve(self, tag, elem, childclass, builtin):
        cls = elem.attrib.get('class', childclass or 'main')
        if cls not in self.classes:
            raise ValueError(f'unknown default class {cls!r}')
        out = dict(builtin)
        out.update(self.classes[cls].get(tag, {}))
        out.update({k: v for k, v in elem.attrib.items() if k != 'class'})
        return out


# ----------------------------------------------------------------------------- mesh loading
def _load_mesh_vertices(path: Path) -> np.ndarray:
    suffix = path.suffix.lower()
    if suffix == '.obj':
        v = [ln.split()[1:4] for ln in open(path, 'r', errors='ignore') if ln.startswith('v ')]
        return np.asarray(v, dtype=np.float64)
    if suffix == '.stl':
        raw = open(path, 'rb').read()
        ntri = int(np.frombuffer(raw[80:84], dtype='<u4')[0])
        if 84 + 50 * ntri == len(raw):  # binary
            rec = np.frombuffer(raw[84:], dtype=np.dtype([('n', '<f4', 3), ('v', '<f4', 9), ('a', '<u2')]), count=ntri)
            return rec['v'].reshape(-1, 3).astype(np.float64)
        v = [ln.split()[1:4] for ln in raw.decode(errors='ignore').splitlines() if ln.strip().startswith('vertex')]
        return np.asarray(v, dtype=np.float64)
    raise ValueError(f'unsupported mesh format {path}')


HULL_MERGE_TOL = 1e-4  # [m] hull vertices closer than this are merged (a tenth of the usual 1 mm contact margin)


def _hull_vertices(v: np.ndarray) -> np.ndarray:
    """Convex-hull vertices of a mesh, with near-duplicate hull vertices (scan artefacts of the OBJ/STL files,
    some only micrometres apart) merged greedily within HULL_MERGE_TOL."""
    from scipy.spatial import ConvexHull, cKDTree

    v = np.unique(np.round(v, 9), axis=0)
    h = v[np.sort(ConvexHull(v).vertices)]
    tree = cKDTree(h)
    keep = np.ones(len(h), dtype=bool)
    for i in range(len(h)):
        if keep[i]:
            for j in tree.query_ball_point(h[i], HULL_MERGE_TOL):
                if j > i:
                    keep[j] = False
    return h[keep]


# ----------------------------------------------------------------------------- model description
@dataclasses.dataclass
class ModelDesc:
    """Flat model tables, MuJoCo naming.  Everything fp64 / int32 numpy."""

    name: str
    # options
    timestep: float
    gravity: np.ndarray
    cone: int  # 0 pyramidal, 1 elliptic
    impratio: float
    integrator: int  # 0 Euler, 3 implicitfast (run as Euler + implicit damping: the same update for torque motors + joint damping)
    # sizes
    nq: int
    nv: int
    nu: int
    nbody: int
    njnt: int
    ngeom: int
    # bodies
    body_names: list
    body_parentid: np.ndarray
    body_pos: np.ndarray
    body_quat: np.ndarray
    body_ipos: np.ndarray
    body_iquat: np.ndarray
    body_mass: np.ndarray
    body_inertia: np.ndarray
    body_jntadr: np.ndarray
    body_jntnum: np.ndarray
    body_dofadr: np.ndarray
    body_invweight0: np.ndarray  # (nbody,2)
    # joints / dofs
    jnt_names: list
    jnt_type: np.ndarray
    jnt_bodyid: np.ndarray
    jnt_qposadr: np.ndarray
    jnt_dofadr: np.ndarray
    jnt_pos: np.ndarray
    jnt_axis: np.ndarray
    jnt_limited: np.ndarray
    jnt_range: np.ndarray
    jnt_margin: np.ndarray
    jnt_solref: np.ndarray
    jnt_solimp: np.ndarray
    jnt_actfrclimited: np.ndarray
    jnt_actfrcrange: np.ndarray
    qpos0: np.ndarray
    dof_bodyid: np.ndarray
    dof_jntid: np.ndarray
    dof_parentid: np.ndarray
    dof_damping: np.ndarray
    dof_armature: np.ndarray
    dof_frictionloss: np.ndarray
    dof_solref: np.ndarray
    dof_solimp: np.ndarray
    dof_invweight0: np.ndarray
    # geoms (collision-enabled and visual alike; visual ones have contype=conaffinity=0)
    geom_names: list
    geom_type: np.ndarray
    geom_bodyid: np.ndarray
    geom_pos: np.ndarray
    geom_quat: np.ndarray
    geom_size: np.ndarray
    geom_friction: np.ndarray
    geom_margin: np.ndarray
    geom_gap: np.ndarray
    geom_condim: np.ndarray
    geom_contype: np.ndarray
    geom_conaffinity: np.ndarray
    geom_priority: np.ndarray
    geom_solref: np.ndarray
    geom_solimp: np.ndarray
    geom_solmix: np.ndarray
    geom_group: np.ndarray
    geom_rbound: np.ndarray
    # collision vertex clouds of robot geoms (shared between geoms that use the same mesh), GEOM frame
    geom_cloudid: np.ndarray  # (ngeom,) -1 = no cloud (visual / world geom)
    cloud_vertadr: np.ndarray
    cloud_vertnum: np.ndarray
    cloud_radius: np.ndarray  # inflation radius (sphere / capsule), 0 for box / mesh
    vert_pos: np.ndarray  # (nvert,3) in the geom's own frame
    # sites
    site_names: list
    site_bodyid: np.ndarray
    site_pos: np.ndarray
    site_quat: np.ndarray
    # actuators
    actuator_names: list
    actuator_trnid: np.ndarray
    actuator_gear: np.ndarray
    actuator_ctrllimited: np.ndarray
    actuator_ctrlrange: np.ndarray
    actuator_forcelimited: np.ndarray
    actuator_forcerange: np.ndarray
    # sensors: (name, type, objname)
    sensors: list
    # keyframes
    key_qpos: np.ndarray
    key_names: list
    # statistics
    meaninertia: float
    total_mass: float
    # <contact><exclude body1= body2=/> pairs (body ids)
    exclude_body1: np.ndarray = dataclasses.field(default_factory=lambda: np.zeros(0, np.int32))
    exclude_body2: np.ndarray = dataclasses.field(default_factory=lambda: np.zeros(0, np.int32))

    # ---- (de)serialisation: plain JSON tables
    def to_json(self) -> str:
        d = {}
        for f in dataclasses.fields(self):
            v = getattr(self, f.name)
            if isinstance(v, np.ndarray):
                d[f.name] = {'dtype': 'i4' if v.dtype.kind == 'i' else 'f8', 'shape': list(v.shape),
                             'data': v.reshape(-1).tolist()}
            else:
                d[f.name] = v
        return json.dumps(d)

    @classmethod
    def from_json(cls, text: str) -> 'ModelDesc':
        d = json.loads(text)
        kw = {}
        for f in dataclasses.fields(cls):
            if f.name not in d:   # field added after the table was written: dataclass default
                continue
            v = d[f.name]
            if isinstance(v, dict) and 'dtype' in v:
                v = np.asarray(v['data'], dtype=np.int32 if v['dtype'] == 'i4' else np.float64).reshape(v['shape'])
            kw[f.name] = v
        return cls(**kw)

    def geom_id(self, name):
        return self.geom_names.index(name)

    def body_id(self, name):
        return self.body_names.index(name)


# ----------------------------------------------------------------------------- the compiler
class _Compiler:
    def __init__(self, xml_path: Path, mesh_hulls=True):
        self.xml_path = Path(xml_path)
        self.mesh_hulls = mesh_hulls
        self.defaults = _Defaults()
        self.meshes: dict[str, dict] = {}
        self.root = self._load_with_includes(self.xml_path)

    def _load_with_includes(self, path: Path):
        root = ET.parse(path).getroot()
        base = path.parent
        for parent in list(root.iter()):
            for i, ch in enumerate(list(parent)):
                if ch.tag == 'include':
                    inc_path = Path(ch.attrib['file'])
                    if not inc_path.is_absolute():
                        inc_path = base / inc_path
                    inc = self._load_with_includes(inc_path)
                    for e in inc.iter():  # make mesh files absolute relative to the included file
                        if e.tag == 'mesh' and 'file' in e.attrib and not os.path.isabs(e.attrib['file']):
                            e.attrib['file'] = str(inc_path.parent / e.attrib['file'])
                    parent.remove(ch)
                    for j, sub in enumerate(list(inc)):
                        parent.insert(i + j, sub)
        return root

    def compile(self) -> ModelDesc:
        root = self.root
        comp = {}
        for c in root.findall('compiler'):
            comp.update(c.attrib)
        if comp.get('angle', 'degree') != 'radian':
            raise ValueError('only <compiler angle="radian"> models are supported')
        self.autolimits = comp.get('autolimits', 'true') == 'true'
        self.meshdir = comp.get('meshdir', '')
        opt = {}
        for o in root.findall('option'):
            opt.update(o.attrib)
        for d in root.findall('default'):
            self.defaults.load(d)
        for a in root.findall('asset'):
            for m in a.findall('mesh'):
                attr = self.defaults.resolve('mesh', m, None, {})
                name = attr.get('name') or Path(attr['file']).stem
                self.meshes[name] = attr

        B = dict(names=['world'], parent=[0], pos=[np.zeros(3)], quat=[np.array([1.0, 0, 0, 0])],
                 ipos=[np.zeros(3)], iquat=[np.array([1.0, 0, 0, 0])], mass=[0.0], inertia=[np.zeros(3)],
                 jntadr=[-1], jntnum=[0], dofadr=[-1])
        J = dict(names=[], type=[], body=[], qposadr=[], dofadr=[], pos=[], axis=[], limited=[], range=[],
                 margin=[], solref=[], solimp=[], afl=[], afr=[], qpos0=[], ref=[])
        D = dict(body=[], jnt=[], parent=[], damping=[], armature=[], frictionloss=[], solref=[], solimp=[])
        G = dict(names=[], type=[], body=[], pos=[], quat=[], size=[], friction=[], margin=[], gap=[], condim=[],
                 contype=[], conaffinity=[], priority=[], solref=[], solimp=[], solmix=[], group=[], mesh=[])
        S = dict(names=[], body=[], pos=[], quat=[])
        self.B, self.J, self.D, self.G, self.S = B, J, D, G, S
        self.nq = self.nv = 0

        for wb in root.findall('worldbody'):
            self._children(wb, 0, None, last_dof=-1)

        # actuators
        A = dict(names=[], trnid=[], gear=[], ctrllimited=[], ctrlrange=[], forcelimited=[], forcerange=[])
        for act in root.findall('actuator'):
            for m in act:
                if m.tag not in ('motor',):
                    raise ValueError(f'actuator type <{m.tag}> not supported (torque motors only)')
                attr = self.defaults.resolve('motor', m, None, _MOTOR_DEF)
                A['names'].append(attr.get('name', f'act{len(A["names"])}'))
                A['trnid'].append(J['names'].index(attr['joint']))
                A['gear'].append(_f(attr['gear'], 6)[0])
                cr = _f(attr['ctrlrange'], 2)
                fr = _f(attr['forcerange'], 2)
                A['ctrlrange'].append(cr)
                A['forcerange'].append(fr)
                A['ctrllimited'].append(self._limited(attr['ctrllimited'], cr))
                A['forcelimited'].append(self._limited(attr['forcelimited'], fr))

        sensors = []
        for sn in root.findall('sensor'):
            for s in sn:
                obj = s.attrib.get('site') or s.attrib.get('joint') or s.attrib.get('objname')
                sensors.append([s.attrib.get('name', ''), s.tag, obj])

        key_qpos, key_names = [], []
        for kf in root.findall('keyframe'):
            for k in kf.findall('key'):
                key_names.append(k.attrib.get('name', ''))
                q = _f(k.attrib['qpos']) if 'qpos' in k.attrib else np.asarray(J['qpos0'], dtype=np.float64)
                key_qpos.append(q)

        nb, nj, ng = len(B['names']), len(J['names']), len(G['names'])
        arr = lambda x, dt=np.float64, shape=None: (np.asarray(x, dtype=dt).reshape(shape) if shape else np.asarray(x, dtype=dt))
        geom_cloudid, cloud_vertadr, cloud_vertnum, cloud_radius, vert_pos, rbound = self._lower_geoms()
        integrator = {'Euler': 0, 'RK4': 1, 'implicit': 2, 'implicitfast': 3}[opt.get('integrator', 'Euler')]
        md = ModelDesc(
            name=root.attrib.get('model', self.xml_path.stem),
            timestep=float(opt.get('timestep', 0.002)), gravity=_f(opt.get('gravity', '0 0 -9.81')),
            cone={'pyramidal': 0, 'elliptic': 1}[opt.get('cone', 'pyramidal')], impratio=float(opt.get('impratio', 1.0)),
            integrator=integrator,
            nq=self.nq, nv=self.nv, nu=len(A['names']), nbody=nb, njnt=nj, ngeom=ng,
            body_names=B['names'], body_parentid=arr(B['parent'], np.int32), body_pos=arr(B['pos']).reshape(nb, 3),
            body_quat=arr(B['quat']).reshape(nb, 4), body_ipos=arr(B['ipos']).reshape(nb, 3),
            body_iquat=arr(B['iquat']).reshape(nb, 4), body_mass=arr(B['mass']), body_inertia=arr(B['inertia']).reshape(nb, 3),
            body_jntadr=arr(B['jntadr'], np.int32), body_jntnum=arr(B['jntnum'], np.int32), body_dofadr=arr(B['dofadr'], np.int32),
            body_invweight0=np.zeros((nb, 2)),
            jnt_names=J['names'], jnt_type=arr(J['type'], np.int32), jnt_bodyid=arr(J['body'], np.int32),
            jnt_qposadr=arr(J['qposadr'], np.int32), jnt_dofadr=arr(J['dofadr'], np.int32),
            jnt_pos=arr(J['pos']).reshape(nj, 3), jnt_axis=arr(J['axis']).reshape(nj, 3),
            jnt_limited=arr(J['limited'], np.int32), jnt_range=arr(J['range']).reshape(nj, 2), jnt_margin=arr(J['margin']),
            jnt_solref=arr(J['solref']).reshape(nj, 2), jnt_solimp=arr(J['solimp']).reshape(nj, 5),
            jnt_actfrclimited=arr(J['afl'], np.int32), jnt_actfrcrange=arr(J['afr']).reshape(nj, 2),
            qpos0=np.concatenate([np.atleast_1d(q) for q in J['qpos0']]) if nj else np.zeros(0),
            dof_bodyid=arr(D['body'], np.int32), dof_jntid=arr(D['jnt'], np.int32), dof_parentid=arr(D['parent'], np.int32),
            dof_damping=arr(D['damping']), dof_armature=arr(D['armature']), dof_frictionloss=arr(D['frictionloss']),
            dof_solref=arr(D['solref']).reshape(self.nv, 2), dof_solimp=arr(D['solimp']).reshape(self.nv, 5),
            dof_invweight0=np.zeros(self.nv),
            geom_names=G['names'], geom_type=arr(G['type'], np.int32), geom_bodyid=arr(G['body'], np.int32),
            geom_pos=arr(G['pos']).reshape(ng, 3), geom_quat=arr(G['quat']).reshape(ng, 4), geom_size=arr(G['size']).reshape(ng, 3),
            geom_friction=arr(G['friction']).reshape(ng, 3), geom_margin=arr(G['margin']), geom_gap=arr(G['gap']),
            geom_condim=arr(G['condim'], np.int32), geom_contype=arr(G['contype'], np.int32),
            geom_conaffinity=arr(G['conaffinity'], np.int32), geom_priority=arr(G['priority'], np.int32),
            geom_solref=arr(G['solref']).reshape(ng, 2), geom_solimp=arr(G['solimp']).reshape(ng, 5),
            geom_solmix=arr(G['solmix']), geom_group=arr(G['group'], np.int32), geom_rbound=rbound,
            geom_cloudid=geom_cloudid, cloud_vertadr=cloud_vertadr, cloud_vertnum=cloud_vertnum, cloud_radius=cloud_radius, vert_pos=vert_pos,
            site_names=S['names'], site_bodyid=arr(S['body'], np.int32), site_pos=arr(S['pos']).reshape(len(S['names']), 3),
            site_quat=arr(S['quat']).reshape(len(S['names']), 4),
            actuator_names=A['names'], actuator_trnid=arr(A['trnid'], np.int32), actuator_gear=arr(A['gear']),
            actuator_ctrllimited=arr(A['ctrllimited'], np.int32), actuator_ctrlrange=arr(A['ctrlrange']).reshape(len(A['names']), 2),
            actuator_forcelimited=arr(A['forcelimited'], np.int32), actuator_forcerange=arr(A['forcerange']).reshape(len(A['names']), 2),
            sensors=sensors,
            key_qpos=arr(key_qpos).reshape(len(key_qpos), self.nq) if key_qpos else np.zeros((0, self.nq)),
            key_names=key_names, meaninertia=0.0, total_mass=float(np.sum(B['mass'])),
            exclude_body1=arr([B['names'].index(e.attrib['body1']) for c in root.findall('contact') for e in c.findall('exclude')], np.int32),
            exclude_body2=arr([B['names'].index(e.attrib['body2']) for c in root.findall('contact') for e in c.findall('exclude')], np.int32),
        )
        set_const(md)
        return md

    def _limited(self, flag, rng):
        if flag == 'auto':
            if not self.autolimits:
                return 0
            return int(rng[0] != 0 or rng[1] != 0)
        return int(flag == 'true')

    def _children(self, elem, body_id, childclass, last_dof):
        """Process geoms/sites of ``elem`` (a body or worldbody) and recurse into child bodies."""
        B, J, D, G, S = self.B, self.J, self.D, self.G, self.S
        for ch in elem:
            if ch.tag == 'geom':
                a = self.defaults.resolve('geom', ch, childclass, _GEOM_DEF)
                gtype = GEOM_MESH if 'mesh' in a else _GEOM_TYPES[a['type']]
                size = _f(a['size'], 3)[:3]
                pos, quat = _f(a['pos'], 3), _f(a['quat'], 4)
                if 'fromto' in a:
                    ft = _f(a['fromto'], 6)
                    p0, p1 = ft[:3], ft[3:]
                    pos = 0.5 * (p0 + p1)
                    quat = quat_z_to_vec(p1 - p0)
                    half = 0.5 * np.linalg.norm(p1 - p0)
                    if gtype in (GEOM_CAPSULE, GEOM_CYLINDER):
                        size = np.array([size[0], half, 0.0])
                    elif gtype == GEOM_BOX:
                        size = np.array([size[0], size[0], half])
                G['names'].append(a.get('name', ''))
                G['type'].append(gtype)
                G['body'].append(body_id)
                G['pos'].append(pos)
                G['quat'].append(quat / np.linalg.norm(quat))
                G['size'].append(size)
                G['friction'].append(_f(a['friction'], 3)[:3] if len(_f(a['friction'])) >= 3 else
                                     np.concatenate([_f(a['friction']), _f(_GEOM_DEF['friction'])[len(_f(a['friction'])):]]))
                G['margin'].append(float(a['margin']))
                G['gap'].append(float(a['gap']))
                G['condim'].append(int(a['condim']))
                G['contype'].append(int(a['contype']))
                G['conaffinity'].append(int(a['conaffinity']))
                G['priority'].append(int(a['priority']))
                G['solref'].append(_f(a['solref'], 2))
                simp = _f(a['solimp'])
                G['solimp'].append(np.concatenate([simp, _f(_GEOM_DEF['solimp'])[len(simp):]]))
                G['solmix'].append(float(a['solmix']))
                G['group'].append(int(a['group']))
                G['mesh'].append(a.get('mesh'))
            elif ch.tag == 'site':
                a = self.defaults.resolve('site', ch, childclass, _SITE_DEF)
                S['names'].append(a.get('name', ''))
                S['body'].append(body_id)
                S['pos'].append(_f(a['pos'], 3))
                S['quat'].append(_f(a['quat'], 4))
        for ch in elem:
            if ch.tag != 'body':
                continue
            cc = ch.attrib.get('childclass', childclass)
            bid = len(B['names'])
            B['names'].append(ch.attrib.get('name', f'body{bid}'))
            B['parent'].append(body_id)
            B['pos'].append(_f(ch.attrib.get('pos', '0 0 0'), 3))
            q = _f(ch.attrib.get('quat', '1 0 0 0'), 4)
            B['quat'].append(q / np.linalg.norm(q))
            inert = ch.find('inertial')
            if inert is None:
                raise ValueError(f'body {B["names"][-1]}: explicit <inertial> required (geom-derived inertia not supported)')
            B['ipos'].append(_f(inert.attrib.get('pos', '0 0 0'), 3))
            iq = _f(inert.attrib.get('quat', '1 0 0 0'), 4)
            B['iquat'].append(iq / np.linalg.norm(iq))
            B['mass'].append(float(inert.attrib['mass']))
            if 'diaginertia' not in inert.attrib:
                raise ValueError('only diaginertia inertials supported')
            B['inertia'].append(_f(inert.attrib['diaginertia'], 3))
            B['jntadr'].append(len(J['names']))
            B['dofadr'].append(self.nv)
            njnt = 0
            ld = last_dof
            for jn in ch:
                if jn.tag not in ('joint', 'freejoint'):
                    continue
                njnt += 1
                jid = len(J['names'])
                if jn.tag == 'freejoint':
                    a = dict(_JOINT_DEF)
                    a.update(type='free')
                    a.update(jn.attrib)
                else:
                    a = self.defaults.resolve('joint', jn, cc, _JOINT_DEF)
                jt = {'free': JNT_FREE, 'ball': JNT_BALL, 'slide': JNT_SLIDE, 'hinge': JNT_HINGE}[a['type']]
                if jt in (JNT_BALL, JNT_SLIDE):
                    raise ValueError('ball/slide joints not supported')
                J['names'].append(a.get('name', f'joint{jid}'))
                J['type'].append(jt)
                J['body'].append(bid)
                J['qposadr'].append(self.nq)
                J['dofadr'].append(self.nv)
                J['pos'].append(_f(a['pos'], 3))
                ax = _f(a['axis'], 3)
                J['axis'].append(ax / np.linalg.norm(ax))
                rng = _f(a['range'], 2)
                J['range'].append(rng)
                J['limited'].append(0 if jt == JNT_FREE else self._limited(a['limited'], rng))
                J['margin'].append(float(a['margin']))
                J['solref'].append(_f(a['solreflimit'], 2))
                J['solimp'].append(_f(a['solimplimit'], 5))
                afr = _f(a['actuatorfrcrange'], 2)
                J['afr'].append(afr)
                J['afl'].append(self._limited(a['actuatorfrclimited'], afr))
                ndof = 6 if jt == JNT_FREE else 1
                if jt == JNT_FREE:
                    J['qpos0'].append(np.concatenate([B['pos'][-1], B['quat'][-1]]))
                    self.nq += 7
                else:
                    J['qpos0'].append(np.array([float(a['ref'])]))
                    self.nq += 1
                for k in range(ndof):
                    D['body'].append(bid)
                    D['jnt'].append(jid)
                    D['parent'].append(ld)
                    ld = self.nv
                    D['damping'].append(0.0 if jt == JNT_FREE else float(a['damping']))
                    D['armature'].append(0.0 if jt == JNT_FREE else float(a['armature']))
                    D['frictionloss'].append(0.0 if jt == JNT_FREE else float(a['frictionloss']))
                    D['solref'].append(_f(a['solreffriction'], 2))
                    D['solimp'].append(_f(a['solimpfriction'], 5))
                    self.nv += 1
            B['jntnum'].append(njnt)
            if njnt == 0:
                B['jntadr'][-1] = -1
                B['dofadr'][-1] = -1
            self._children(ch, bid, cc, ld)

    def _lower_geoms(self):
        """Vertex-cloud lowering of every collision-enabled geom that is attached to a moving body."""
        G = self.G
        ng = len(G['names'])
        cloudid = np.full(ng, -1, np.int32)
        rbound = np.zeros(ng)
        clouds, radii, keys = [], [], {}
        for g in range(ng):
            if G['body'][g] == 0 or (G['contype'][g] == 0 and G['conaffinity'][g] == 0):
                continue
            t, size = G['type'][g], G['size'][g]
            if t == GEOM_SPHERE:
                key, loc, r = ('s', size[0]), np.zeros((1, 3)), size[0]
            elif t == GEOM_CAPSULE:
                key, loc, r = ('c', size[0], size[1]), np.array([[0, 0, -size[1]], [0, 0, size[1]]]), size[0]
            elif t == GEOM_BOX:
                key = ('b',) + tuple(size)
                # corner i = (x sign: bit 0, y sign: bit 1, z sign: bit 2): the order mjraw_PlaneBox walks the corners in
                loc = np.array([[sx * size[0], sy * size[1], sz * size[2]] for sz in (-1, 1) for sy in (-1, 1) for sx in (-1, 1)])
                r = 0.0
            elif t == GEOM_CYLINDER:
                key = ('y', size[0], size[1])
                ang = np.arange(16) * (2 * np.pi / 16)
                ring = np.stack([size[0] * np.cos(ang), size[0] * np.sin(ang)], 1)
                loc = np.concatenate([np.c_[ring, np.full(16, -size[1])], np.c_[ring, np.full(16, size[1])]])
                r = 0.0
            elif t == GEOM_MESH:
                if not self.mesh_hulls:
                    continue
                key = ('m', G['mesh'][g])
                loc, r = None, 0.0
                if key not in keys:
                    m = self.meshes[G['mesh'][g]]
                    fpath = Path(m['file'])
                    if not fpath.is_absolute():
                        fpath = self.xml_path.parent / self.meshdir / fpath
                    if not fpath.exists():
                        continue  # mesh blob missing from the checkout (.MISSING_LARGE_BLOBS)
                    loc = _hull_vertices(_load_mesh_vertices(fpath) * _f(m.get('scale', '1 1 1'), 3))
            else:
                continue
            if key not in keys:
                keys[key] = len(clouds)
                clouds.append(np.asarray(loc, dtype=np.float64))
                radii.append(float(r))
            cloudid[g] = keys[key]
            c = clouds[keys[key]]
            rbound[g] = np.max(np.linalg.norm(c, axis=1)) + radii[keys[key]]
        vertnum = np.array([len(c) for c in clouds], np.int32)
        vertadr = np.concatenate([[0], np.cumsum(vertnum)[:-1]]).astype(np.int32) if len(clouds) else np.zeros(0, np.int32)
        vp = np.concatenate(clouds) if clouds else np.zeros((0, 3))
        return cloudid, vertadr, vertnum, np.asarray(radii, dtype=np.float64), vp, rbound


# ----------------------------------------------------------------------------- compile-time constants
def kinematics_qpos0(md: ModelDesc, qpos=None):
    """World poses of all bodies at ``qpos`` (default qpos0). Returns xpos(nb,3), xmat(nb,3,3), xquat(nb,4)."""
    qpos = md.qpos0 if qpos is None else qpos
    nb = md.nbody
    xpos, xquat = np.zeros((nb, 3)), np.zeros((nb, 4))
    xquat[0] = [1, 0, 0, 0]
    for b in range(1, nb):
        p = md.body_parentid[b]
        Rp = quat_to_mat(xquat[p])
        pos = xpos[p] + Rp @ md.body_pos[b]
        quat = quat_mul(xquat[p], md.body_quat[b])
        for j in range(md.body_jntadr[b], md.body_jntadr[b] + md.body_jntnum[b]) if md.body_jntnum[b] else []:
            qa = md.jnt_qposadr[j]
            if md.jnt_type[j] == JNT_FREE:
                pos = qpos[qa:qa + 3].copy()
                quat = qpos[qa + 3:qa + 7] / np.linalg.norm(qpos[qa + 3:qa + 7])
            else:
                quat = quat_mul(quat, axis_angle_quat(md.jnt_axis[j], qpos[qa] - md.qpos0[qa]))
        xpos[b], xquat[b] = pos, quat
    xmat = np.stack([quat_to_mat(q) for q in xquat])
    return xpos, xmat, xquat


def _dof_axes(md: ModelDesc, xpos, xmat):
    """Per-dof (angular axis, linear axis, anchor) in world coordinates. Linear dofs have zero angular axis."""
    ang, lin, anchor = np.zeros((md.nv, 3)), np.zeros((md.nv, 3)), np.zeros((md.nv, 3))
    for j in range(md.njnt):
        b, d = md.jnt_bodyid[j], md.jnt_dofadr[j]
        if md.jnt_type[j] == JNT_FREE:
            for k in range(3):
                lin[d + k, k] = 1.0
                ang[d + 3 + k] = xmat[b][:, k]
                anchor[d + 3 + k] = xpos[b]
        else:
            ang[d] = xmat[b] @ md.jnt_axis[j]
            anchor[d] = xpos[b] + xmat[b] @ md.jnt_pos[j]
    return ang, lin, anchor


def point_jacobian(md: ModelDesc, xpos, xmat, point, body):
    """(jacp, jacr) 3 x nv of a world point rigidly attached to ``body`` (semantics of ``mujoco.mj_jac``)."""
    ang, lin, anchor = _dof_axes(md, xpos, xmat)
    jacp, jacr = np.zeros((3, md.nv)), np.zeros((3, md.nv))
    chain = set()
    b = body
    while b > 0:
        if md.body_jntnum[b]:
            for j in range(md.body_jntadr[b], md.body_jntadr[b] + md.body_jntnum[b]):
                n = 6 if md.jnt_type[j] == JNT_FREE else 1
                chain.update(range(md.jnt_dofadr[j], md.jnt_dofadr[j] + n))
        b = md.body_parentid[b]
    for d in chain:
        jacr[:, d] = ang[d]
        jacp[:, d] = lin[d] + np.cross(ang[d], point - anchor[d])
    return jacp, jacr


def mass_matrix_dense(md: ModelDesc, qpos=None):
    """Joint-space inertia via sum_b J_b^T I_b J_b (+ armature). Independent of the CRBA used by the oracle/GPU."""
    xpos, xmat, _ = kinematics_qpos0(md, qpos)
    M = np.diag(md.dof_armature.astype(np.float64))
    for b in range(1, md.nbody):
        com = xpos[b] + xmat[b] @ md.body_ipos[b]
        Ri = xmat[b] @ quat_to_mat(md.body_iquat[b])
        Iw = Ri @ np.diag(md.body_inertia[b]) @ Ri.T
        jp, jr = point_jacobian(md, xpos, xmat, com, b)
        M += md.body_mass[b] * jp.T @ jp + jr.T @ Iw @ jr
    return M, xpos, xmat


def set_const(md: ModelDesc):
    """dof_invweight0 / body_invweight0 / meaninertia at qpos0 (MuJoCo engine_setconst.c ``set0`` semantics)."""
    M, xpos, xmat = mass_matrix_dense(md)
    Minv = np.linalg.inv(M)
    md.meaninertia = float(np.mean(np.diag(M))) if md.nv else 1.0
    # dof_invweight0: diagonal of M^-1, averaged over the 3 translational / 3 rotational dofs of a free joint
    inv = np.diag(Minv).copy()
    for j in range(md.njnt):
        d = md.jnt_dofadr[j]
        if md.jnt_type[j] == JNT_FREE:
            inv[d:d + 3] = np.mean(inv[d:d + 3])
            inv[d + 3:d + 6] = np.mean(inv[d + 3:d + 6])
    md.dof_invweight0 = inv
    # body_invweight0: mean diagonal of the translational / rotational blocks of J M^-1 J^T at the body com
    biw = np.zeros((md.nbody, 2))
    for b in range(1, md.nbody):
        com = xpos[b] + xmat[b] @ md.body_ipos[b]
        jp, jr = point_jacobian(md, xpos, xmat, com, b)
        Jb = np.vstack([jp, jr])
        A = Jb @ Minv @ Jb.T
        biw[b, 0] = max(MJ_MINVAL, np.trace(A[:3, :3]) / 3)
        biw[b, 1] = max(MJ_MINVAL, np.trace(A[3:, 3:]) / 3)
    md.body_invweight0 = biw


CLOUD_CHUNK = 64   # = one wavefront: the kernels scan a cloud in chunks of this many consecutive vertices


def sort_cloud_vertices(md: 'ModelDesc') -> 'ModelDesc':
    """Order the vertices of every hull cloud larger than one chunk along the cloud's principal axis (stable, idempotent), so
    that a chunk of CLOUD_CHUNK consecutive vertices is a compact slice of the hull: the kernels bound each chunk by its box
    and skip the chunks that cannot reach the world geom under test (csrc/gq_boxes.h).  The SET of vertices - all that the
    deepest-vertex rule depends on - is unchanged; kernel and oracle read the same (sorted) table."""
    vp = np.array(md.vert_pos, dtype=np.float64)
    for c in range(len(md.cloud_vertnum)):
        a, n = int(md.cloud_vertadr[c]), int(md.cloud_vertnum[c])
        if n <= CLOUD_CHUNK:
            continue
        v = vp[a:a + n]
        _, _, vt = np.linalg.svd(v - v.mean(0), full_matrices=False)
        ax = vt[0] * (1.0 if vt[0][np.argmax(np.abs(vt[0]))] > 0 else -1.0)   # sign fixed: deterministic
        vp[a:a + n] = v[np.argsort(np.round(v @ ax, 9), kind='stable')]
    md.vert_pos = vp
    return md


def compile_mjcf(xml_path, mesh_hulls=True) -> ModelDesc:
    """Compile an MJCF file (robot alone, or a scene that <include>s the robot) into a :class:`ModelDesc`."""
    return sort_cloud_vertices(_Compiler(Path(xml_path), mesh_hulls=mesh_hulls).compile())


_MODEL_DIR = Path(__file__).parent / 'model_data'


def load_compiled(robot_file_stem: str) -> ModelDesc:
    """Load a committed, pre-compiled robot table (``model_data/<stem>.json``)."""
    p = _MODEL_DIR / f'{robot_file_stem}.json'
    if not p.exists():
        raise FileNotFoundError(f'no compiled model table {p}; run tools/compile_models.py or pass mjcf_path=')
    return sort_cloud_vertices(ModelDesc.from_json(p.read_text()))
