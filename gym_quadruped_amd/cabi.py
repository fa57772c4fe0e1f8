"""ctypes mirror of ``include/gq.h`` (the C-ABI boundary) and the ModelDesc -> GqModelDesc marshaller.

The reference reaches its physics through the ``mujoco`` pybind API (SURVEY.md §8b lower boundary); this
module is the equivalent binding layer for ``libgq.so``: plain pointers and sizes, no torch types.
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from .mjcf import ModelDesc

GQ_NLEG = 4
GQ_ABI_VERSION = 630   # include/gq.h
# optional extra output rows of the step kernel (include/gq.h gq_batch_set_outputs)
GQ_DYN = dict(MC=0, MB=108, BIAS=144, XPOS=162, XMAT=201, FOOT=318, STRIDE=336)
GQ_CON_MAX, GQ_CON_REC = 12, 24
GQ_CON_STRIDE = 8 + GQ_CON_MAX * GQ_CON_REC

_I = C.POINTER(C.c_int32)
_D = C.POINTER(C.c_double)


class GqModelDesc(C.Structure):
    _fields_ = [
        ('struct_size', C.c_int32),
        ('nq', C.c_int32), ('nv', C.c_int32), ('nu', C.c_int32), ('nbody', C.c_int32), ('njnt', C.c_int32),
        ('ngeom', C.c_int32), ('ncloud', C.c_int32), ('nvert', C.c_int32),
        ('timestep', C.c_double), ('gravity', C.c_double * 3), ('cone', C.c_int32), ('impratio', C.c_double),
        ('integrator', C.c_int32),
        ('body_parentid', _I), ('body_pos', _D), ('body_quat', _D), ('body_ipos', _D), ('body_iquat', _D),
        ('body_mass', _D), ('body_inertia', _D), ('body_jntadr', _I), ('body_jntnum', _I), ('body_invweight0', _D),
        ('jnt_type', _I), ('jnt_bodyid', _I), ('jnt_qposadr', _I), ('jnt_dofadr', _I), ('jnt_pos', _D),
        ('jnt_axis', _D), ('jnt_limited', _I), ('jnt_range', _D), ('jnt_margin', _D), ('jnt_solref', _D),
        ('jnt_solimp', _D), ('jnt_actfrclimited', _I), ('jnt_actfrcrange', _D), ('qpos0', _D),
        ('dof_bodyid', _I), ('dof_jntid', _I), ('dof_parentid', _I), ('dof_damping', _D), ('dof_armature', _D),
        ('dof_frictionloss', _D), ('dof_solref', _D), ('dof_solimp', _D), ('dof_invweight0', _D),
        ('geom_bodyid', _I), ('geom_pos', _D), ('geom_quat', _D), ('geom_cloudid', _I), ('geom_friction', _D),
        ('geom_margin', _D), ('geom_gap', _D), ('geom_condim', _I), ('geom_priority', _I), ('geom_solref', _D),
        ('geom_solimp', _D), ('geom_solmix', _D), ('geom_rbound', _D),
        ('cloud_vertadr', _I), ('cloud_vertnum', _I), ('cloud_radius', _D), ('vert_pos', _D),
        ('actuator_trnid', _I), ('actuator_gear', _D), ('actuator_ctrllimited', _I), ('actuator_ctrlrange', _D),
        ('actuator_forcelimited', _I), ('actuator_forcerange', _D),
        ('floor_friction', C.c_double * 3), ('floor_margin', C.c_double), ('floor_gap', C.c_double),
        ('floor_solmix', C.c_double), ('floor_solref', C.c_double * 2), ('floor_solimp', C.c_double * 5),
        ('floor_condim', C.c_int32), ('floor_priority', C.c_int32),
        ('feet_geomid', C.c_int32 * GQ_NLEG), ('terrain_limits', C.c_double * 4), ('meaninertia', C.c_double),
        ('key_qpos', C.c_double * 19),
        ('solver', C.c_int32), ('iterations', C.c_int32), ('tolerance', C.c_double), ('noise_floor', C.c_double),
        ('nbox', C.c_int32), ('box_pos', _D), ('box_mat', _D), ('box_size', _D), ('box_friction', _D), ('box_margin', _D),
        ('box_gap', _D), ('box_solmix', _D), ('box_solref', _D), ('box_solimp', _D), ('box_condim', _I), ('box_priority', _I),
        ('hfield_nrow', C.c_int32), ('hfield_ncol', C.c_int32), ('hfield_data', C.POINTER(C.c_float)),
        ('hfield_size', C.c_double * 4), ('hfield_pos', C.c_double * 3),
        ('hfield_friction', C.c_double * 3), ('hfield_margin', C.c_double), ('hfield_gap', C.c_double),
        ('hfield_solmix', C.c_double), ('hfield_solref', C.c_double * 2), ('hfield_solimp', C.c_double * 5),
        ('hfield_condim', C.c_int32), ('hfield_priority', C.c_int32),
        ('nselfpair', C.c_int32), ('selfpair_geom1', _I), ('selfpair_geom2', _I), ('geom_capsule', _D), ('geom_type', _I),
        ('plane_grid', C.c_int32), ('plane_vert_pos', _D), ('plane_mask', _I),
        ('nadj', C.c_int32), ('vert_adjadr', _I), ('vert_adjnum', _I), ('vert_adj', _I), ('plane_order', _I), ('plane_cap', _D), ('support_grid', _D), ('self_convex', C.c_int32),
    ]


class GqState(C.Structure):
    _fields_ = [('qpos', C.c_void_p), ('qvel', C.c_void_p), ('qacc', C.c_void_p), ('qacc_warmstart', C.c_void_p),
                ('qfrc_applied', C.c_void_p), ('time', C.c_void_p), ('friction', C.c_void_p), ('cmd', C.c_void_p)]


class GqResetCfg(C.Structure):
    _fields_ = [('seed', C.c_uint64), ('random', C.c_int32), ('q_pos_amp', C.c_float), ('q_vel_amp', C.c_float),
                ('roll_sweep', C.c_float), ('pitch_sweep', C.c_float), ('hip_height', C.c_float),
                ('lin_vel_range', C.c_float * 2), ('ang_vel_range', C.c_float * 2), ('friction_range', C.c_float * 2),
                ('cmd_forward', C.c_int32), ('cmd_random', C.c_int32), ('cmd_rotate', C.c_int32), ('cmd_human', C.c_int32),
                ('env_id_offset', C.c_int32), ('autoreset_next_step', C.c_int32)]


class GqImuCfg(C.Structure):
    _fields_ = [('site_pos', C.c_double * 3), ('site_quat', C.c_double * 4), ('accel_noise', C.c_float),
                ('gyro_noise', C.c_float), ('accel_bias_rate', C.c_float), ('gyro_bias_rate', C.c_float), ('seed', C.c_uint64)]


class GqObsOut(C.Structure):
    _fields_ = [('obs', C.c_void_p), ('reward', C.c_void_p), ('terminated', C.c_void_p), ('truncated', C.c_void_p),
                ('invalid_contact', C.c_void_p), ('step_num', C.c_void_p), ('step_num_prev', C.c_void_p), ('contacts_dropped', C.c_void_p)]


class GqPolicyPd(C.Structure):
    _fields_ = [('kp', C.c_float * 12), ('kd', C.c_float * 12), ('q_des', C.c_float * 12), ('noise_sigma', C.c_float), ('noise_seed', C.c_uint64),
                ('noise_step0', C.c_int32)]


class GqMailboxView(C.Structure):
    _fields_ = [('action', C.c_void_p), ('steps_done', C.c_void_p), ('queue_items', C.c_void_p), ('queue_counters', C.c_void_p), ('status', C.c_void_p),
                ('n_queues', C.c_int32), ('queue_capacity', C.c_int32), ('counter_stride', C.c_int32), ('xcc_queue', C.c_int32 * 16)]


class GqResampleCfg(C.Structure):
    _fields_ = [('seed', C.c_uint64), ('cmd_reset', C.c_int32), ('dist_reset', C.c_int32), ('dist_kind', C.c_int32 * 6),
                ('dist_range', (C.c_float * 2) * 6), ('env_id_offset', C.c_int32)]


# index into QuadrupedEnv.ALL_OBS (reference quadruped_env.py:35-66,81) == enum GqObsId
ALL_OBS = [
    'base_pos', 'base_lin_vel', 'base_lin_vel_err', 'base_lin_acc', 'base_ang_vel', 'base_ang_vel_err',
    'base_ori_euler_xyz', 'base_ori_quat_wxyz', 'base_ori_SO3', 'gravity_vector:base',
    'base_lin_vel:base', 'base_lin_vel_err:base', 'base_lin_acc:base', 'base_ang_vel:base', 'base_ang_vel_err:base',
    'qpos', 'qvel', 'tau_ctrl_setpoint', 'qpos_js', 'qvel_js', 'kinetic_energy', 'work',
    'feet_pos', 'feet_pos:base', 'feet_vel', 'feet_vel_rel', 'feet_vel:base', 'feet_vel_rel:base',
    'contact_state', 'contact_forces', 'contact_forces:base',
]
# sensor observables appended after ALL_OBS (enum GqObsId ids 31..36; reference sensors/imu.py:17-18)
IMU_OBS = ['imu_acc', 'imu_acc_noise', 'imu_acc_bias', 'imu_gyro', 'imu_gyro_noise', 'imu_gyro_bias']
OBS_NAMES = ALL_OBS + IMU_OBS
OBS_DIMS = [3, 3, 3, 3, 3, 3, 3, 4, 9, 3, 3, 3, 3, 3, 3, 19, 18, 12, 12, 12, 1, 1, 12, 12, 12, 12, 12, 12, 4, 12, 12] + [3] * 6
LEG_NAMES = ['FL', 'FR', 'RL', 'RR']

SOLVER_PGS, SOLVER_NEWTON = 0, 1


PLANE_GRID = 8   # cube-map cells per face edge of the plane-support tables (6 * 8 * 8 = 384 direction cells)


def plane_cell_tables(grid=PLANE_GRID):
    """Unit centre and angular radius of every cube-map cell (face = dominant axis and its sign, then a grid x grid raster over the
    other two coordinates divided by the dominant one): cell index = ((axis * 2 + negative) * grid + iu) * grid + iv - the
    formula the kernel evaluates (csrc/gq_step_body.h stage_collision_scan)."""
    cen, rad = np.zeros((6 * grid * grid, 3)), np.zeros(6 * grid * grid)
    for m in range(3):
        o = [k for k in range(3) if k != m]
        for sgn in range(2):
            for iu in range(grid):
                for iv in range(grid):
                    pts = []
                    for du in (0.0, 0.5, 1.0):
                        for dv in (0.0, 0.5, 1.0):
                            q = np.zeros(3)
                            q[m] = 1.0 if sgn == 0 else -1.0
                            q[o[0]], q[o[1]] = (iu + du) / grid * 2 - 1, (iv + dv) / grid * 2 - 1
                            pts.append(q / np.linalg.norm(q))
                    pts = np.asarray(pts)
                    idx = ((m * 2 + sgn) * grid + iu) * grid + iv
                    cen[idx] = pts[4]
                    rad[idx] = np.arccos(np.clip((pts @ pts[4]).min(), -1.0, 1.0))   # the corners are the farthest points of a cell
    return cen, rad


def plane_cell_of(d, grid=PLANE_GRID):
    """Cell of direction d (numpy restatement of the kernel's formula; tests)."""
    a = np.abs(d)
    m = 0 if (a[0] >= a[1] and a[0] >= a[2]) else (1 if a[1] >= a[2] else 2)
    o = [k for k in range(3) if k != m]
    iu = min(int((d[o[0]] / a[m] + 1.0) * 0.5 * grid), grid - 1)
    iv = min(int((d[o[1]] / a[m] + 1.0) * 0.5 * grid), grid - 1)
    return ((m * 2 + (0 if d[m] > 0 else 1)) * grid + iu) * grid + iv


def _clip(poly, a, b, c):
    """Sutherland-Hodgman: the part of the convex polygon poly [(u, v)] with a + b u + c v >= 0."""
    out = []
    n = len(poly)
    for i in range(n):
        p, q = poly[i], poly[(i + 1) % n]
        fp, fq = a + b * p[0] + c * p[1], a + b * q[0] + c * q[1]
        if fp >= 0:
            out.append(p)
        if (fp >= 0) != (fq >= 0):
            t = fp / (fp - fq)
            out.append((p[0] + t * (q[0] - p[0]), p[1] + t * (q[1] - p[1])))
    return out


def _cone_cells(V, hull, adj, grid, slack):
    """[cells][vertices] bool: the cube-map cells the normal cone of every hull vertex reaches.  Vertex v supports direction d iff
    (V[v] - V[w]) . d >= 0 for every hull neighbour w; on a cube face (dominant coordinate +-1, the other two = (u, v)) these are
    half-planes in (u, v), the cone's trace is their intersection with the face square - a small convex polygon - and the cells its
    bounding box overlaps are marked (a superset of the cells it meets).  A vertex qhull did not report (no facets) reaches every cell."""
    n = len(V)
    nb = [set() for _ in range(n)]
    for simp in hull.simplices:
        for i in simp:
            nb[i].update(int(j) for j in simp if j != i)
    out = np.zeros((6 * grid * grid, n), dtype=bool)
    scale = np.abs(V).max() + 1e-12
    for v in range(n):
        if not adj[v]:
            out[:, v] = True
            continue
        E = (V[v][None, :] - V[sorted(nb[v])]) / scale            # rows e: e . d >= 0
        for m in range(3):
            o = [k for k in range(3) if k != m]
            for sgn in range(2):
                dm = 1.0 if sgn == 0 else -1.0
                poly = [(-1.0, -1.0), (1.0, -1.0), (1.0, 1.0), (-1.0, 1.0)]
                for e in E:
                    poly = _clip(poly, e[m] * dm + 1e-9, e[o[0]], e[o[1]])   # (+1e-9: the cone a hair wider - conservative)
                    if not poly:
                        break
                if not poly:
                    continue
                P = np.asarray(poly)
                lo, hi = P.min(0) - slack, P.max(0) + slack
                iu0, iu1 = max(int(np.floor((lo[0] + 1) * 0.5 * grid)), 0), min(int(np.floor((hi[0] + 1) * 0.5 * grid)), grid - 1)
                iv0, iv1 = max(int(np.floor((lo[1] + 1) * 0.5 * grid)), 0), min(int(np.floor((hi[1] + 1) * 0.5 * grid)), grid - 1)
                base = (m * 2 + sgn) * grid
                for iu in range(iu0, iu1 + 1):
                    out[(base + iu) * grid + iv0:(base + iu) * grid + iv1 + 1, v] = True
    return out


_PLANE_CACHE = {}


def plane_support_tables(md: ModelDesc, grid=PLANE_GRID, chunk=64, wide_deg=45.0, slack=2e-3):
    key = (hash(np.asarray(md.vert_pos, dtype=np.float64).tobytes()), tuple(int(x) for x in md.cloud_vertnum), grid, chunk, wide_deg, slack)
    if key not in _PLANE_CACHE:
        _PLANE_CACHE[key] = _plane_support_tables(md, grid, chunk, wide_deg, slack)
    return _PLANE_CACHE[key]


SUPPORT_GRID = 16   # cells per edge of a cube-map face (include/gq.h GQ_SUPPORT_GRID: 17 x 17 nodes per face; 32 x 32 was measured: within the noise of 16 x 16)


def support_grid_nodes(grid=SUPPORT_GRID):
    """The 6 * (grid + 1)^2 node directions of GqModelDesc.support_grid, NOT normalised: face f of +x, -x, +y, -y, +z, -z, node (i, j)."""
    t = -1.0 + 2.0 * np.arange(grid + 1) / grid
    U = np.zeros((6, grid + 1, grid + 1, 3))
    for f in range(6):
        major, sign = f // 2, 1.0 - 2.0 * (f % 2)
        oa, ob = [k for k in range(3) if k != major]
        U[f, :, :, major] = sign
        U[f, :, :, oa] = t[:, None]
        U[f, :, :, ob] = t[None, :]
    return U


def support_grids(md: ModelDesc, grid=SUPPORT_GRID):
    """GqModelDesc.support_grid: per cloud its support function at the cube-map nodes, rounded UP (the kernel evaluates the bilinear blend in
    fp32: a relative 1e-6 and an absolute micrometre on top keep the blend an upper bound).  [ncloud][6][grid+1][grid+1] float64."""
    V = np.asarray(md.vert_pos, dtype=np.float64)
    U = support_grid_nodes(grid).reshape(-1, 3)
    out = np.zeros((len(md.cloud_vertnum), 6, grid + 1, grid + 1))
    for cl in range(len(md.cloud_vertnum)):
        n, a = int(md.cloud_vertnum[cl]), int(md.cloud_vertadr[cl])
        if n == 0:
            continue
        h = (V[a:a + n] @ U.T).max(0)
        out[cl] = (h + 1e-6 * np.abs(h) + 1e-6).reshape(6, grid + 1, grid + 1)
    return out


def _patch_order(ax, ids, chunk):
    """Order the vertices ``ids`` so that every block of ``chunk`` consecutive ones is a compact PATCH of directions (ax: the vertices' mean
    outward normals): recursive bisection along the principal axis of the normals, the left part taking a whole number of chunks.  (Sorting
    by cube-map cell, rounds 4 - 5, made row-major STRIPS: their caps - csrc/gq_convex.h picks the chunks of a support query by them - had
    half-angles of 120 - 170 degrees and let 5 of a hull's 8 chunks through; patches let 2 through.)"""
    ids = np.asarray(ids, dtype=np.int64)
    nleaf = (len(ids) + chunk - 1) // chunk
    if nleaf <= 1:
        return ids
    k1 = nleaf // 2
    X = ax[ids]
    _, _, vt = np.linalg.svd(X - X.mean(0), full_matrices=False)
    o = np.argsort(X @ vt[0], kind='stable')
    return np.concatenate([_patch_order(ax, ids[o[:chunk * k1]], chunk), _patch_order(ax, ids[o[chunk * k1:]], chunk)])


def _plane_support_tables(md: ModelDesc, grid, chunk, wide_deg, slack):
    """Acceleration structure of the hull-versus-plane narrow phase (mjc_PlaneConvex's support vertex = the vertex deepest along the
    plane normal): per hull cloud of more than one 64-vertex chunk, (i) its vertices in DIRECTION order - sorted by the cube-map cell
    of the mean of the outward normals of the hull facets around the vertex, vertices with a wide normal cone last - so that a chunk
    of consecutive vertices answers a patch of directions, and (ii) per direction cell a bit mask of the chunks that can hold the
    support vertex of SOME direction in the cell.  A vertex is the support vertex for exactly the directions inside the cone spanned
    by the normals of its facets; the mask keeps a chunk when the cap around one of its vertices' cones (axis = mean normal,
    half-angle = the widest facet normal) comes within the cell's own cap - a superset of the exact answer, so the kernel's scan of
    the masked chunks finds the same vertex as a scan of the whole cloud (tests/test_host_and_abi.py).  Position-sorted chunks
    (sort_cloud_vertices, used against world boxes and the height field) cannot answer this query for a link LYING on the floor: every
    slab along its axis is equally deep, all 8 - 11 chunks of every geom were scanned - 18 k cycles for exactly the waves that end a
    launch.  Returns (plane_vert_pos [nvert][3] f64, plane_mask [ncloud][6 grid^2] int32)."""
    from scipy.spatial import ConvexHull
    cen, rad = plane_cell_tables(grid)
    vert = np.array(md.vert_pos, dtype=np.float64, copy=True)
    ncl = len(md.cloud_vertnum)
    masks = np.ones((ncl, 6 * grid * grid), dtype=np.int32)
    perm = np.zeros(len(vert), dtype=np.int32)
    caps = np.zeros((ncl, 16, 4)); caps[:, :, 3] = -2.0
    for cl in range(ncl):
        n, a = int(md.cloud_vertnum[cl]), int(md.cloud_vertadr[cl])
        perm[a:a + n] = np.arange(n)
        if n <= chunk:
            continue
        V = vert[a:a + n].copy()
        try:
            hull = ConvexHull(V)
        except Exception:   # a degenerate cloud: every chunk is scanned, the order stays
            masks[cl] = (1 << ((n + chunk - 1) // chunk)) - 1
            continue
        N = hull.equations[:, :3]
        adj = [[] for _ in range(n)]
        for f, simp in enumerate(hull.simplices):
            for v in simp:
                adj[v].append(f)
        ax, th = np.zeros((n, 3)), np.full(n, np.pi)
        ctr = V.mean(0)
        for v in range(n):
            if adj[v]:
                mean = N[adj[v]].sum(0)
                nrm = np.linalg.norm(mean)
                if nrm > 1e-9:
                    ax[v] = mean / nrm
                    th[v] = np.arccos(np.clip((N[adj[v]] @ ax[v]).min(), -1.0, 1.0))
                    continue
            # not reported as a hull vertex (qhull merged it into a facet) or a degenerate fan: may be the support vertex of any direction
            out = V[v] - ctr
            ax[v] = out / max(np.linalg.norm(out), 1e-12)
        wide = th > np.radians(wide_deg)
        order = np.concatenate([_patch_order(ax, np.nonzero(~wide)[0], chunk), np.nonzero(wide)[0]]).astype(np.int64)
        vert[a:a + n] = V[order]
        perm[a:a + n] = order
        chunk_of = np.empty(n, dtype=np.int64)
        chunk_of[order] = np.arange(n) // chunk
        ang = np.arccos(np.clip(cen @ ax.T, -1.0, 1.0))                    # [cells][vertices]
        touch = ang <= th[None, :] + rad[:, None] + slack
        touch &= _cone_cells(V, hull, adj, grid, slack)                     # two supersets of the exact answer: so is their intersection
        m = np.zeros(6 * grid * grid, dtype=np.int64)
        for k in range((n + chunk - 1) // chunk):
            m |= (touch[:, chunk_of == k].any(1).astype(np.int64) << k)
        masks[cl] = m.astype(np.int32)
        for k in range((n + chunk - 1) // chunk):
            # the chunk holds the support vertex of a direction only if the direction lies in the normal cone of one of its vertices, the
            # spherical hull of the normals of the vertex's facets: a cap of less than a hemisphere around all those facet normals holds
            # every such cone (such a cap is geodesically convex).  Axis: a few steps towards the smallest enclosing cap.
            mem = np.nonzero(chunk_of == k)[0]
            if np.any(th[mem] >= np.pi - 1e-9):
                continue   # a vertex that may support any direction: the chunk is always scanned
            Nk = N[np.unique(np.concatenate([adj[v] for v in mem]))]
            a_k = Nk.sum(0)
            if np.linalg.norm(a_k) < 1e-9:
                continue
            a_k /= np.linalg.norm(a_k)
            for it in range(1, 65):
                far = Nk[np.argmin(Nk @ a_k)]
                a_k = a_k + (far - a_k) / (it + 1.0)
                a_k /= np.linalg.norm(a_k)
            half = float(np.arccos(np.clip((Nk @ a_k).min(), -1.0, 1.0))) + slack
            if half < np.radians(88.0):
                caps[cl, k, :3], caps[cl, k, 3] = a_k, np.cos(half)
                continue
            # wider than that: every vertex's cone lies within th of its mean normal, whatever the angles (triangle inequality)
            a_k = ax[mem].sum(0)
            if np.linalg.norm(a_k) < 1e-9:
                continue
            a_k /= np.linalg.norm(a_k)
            half = float(np.max(np.arccos(np.clip(ax[mem] @ a_k, -1.0, 1.0)) + th[mem])) + slack
            if half < np.pi:
                caps[cl, k, :3], caps[cl, k, 3] = a_k, np.cos(half)
    return vert, masks, perm, caps


_GRAPH_CACHE = {}
GEOM_MESH = 7


def hull_graphs_enabled(md: ModelDesc) -> bool:
    return any(int(md.geom_cloudid[g]) >= 0 and int(md.geom_type[g]) == GEOM_MESH and int(md.geom_bodyid[g]) > 0 for g in range(md.ngeom))


def hull_graphs(md: ModelDesc):
    """Edge graphs of the mesh clouds' convex hulls (MuJoCo's ``mesh_graph``, which mjc_PlaneConvex walks: quadruped_env.py:271 ->
    mj_step -> mj_collision): per vertex of ``vert_pos`` the cloud-local indices of the vertices it shares a hull edge with, ascending.
    Returns (vert_adjadr [nvert], vert_adjnum [nvert], vert_adj [nadj]) int32; vertices of clouds no mesh geom uses have no neighbours.
    The hull is scipy's (qhull, triangulated facets), so a planar polygonal facet contributes its fan's diagonals as edges too."""
    key = (hash(np.asarray(md.vert_pos, dtype=np.float64).tobytes()), tuple(int(x) for x in md.cloud_vertnum), tuple(int(x) for x in md.geom_type),
           tuple(int(x) for x in md.geom_cloudid))
    if key in _GRAPH_CACHE:
        return _GRAPH_CACHE[key]
    from scipy.spatial import ConvexHull
    nvert = int(md.vert_pos.shape[0])
    adr, num, adj = np.zeros(nvert, np.int32), np.zeros(nvert, np.int32), []
    mesh_clouds = {int(md.geom_cloudid[g]) for g in range(md.ngeom) if int(md.geom_cloudid[g]) >= 0 and int(md.geom_type[g]) == GEOM_MESH}
    for cl in sorted(mesh_clouds):
        n, a = int(md.cloud_vertnum[cl]), int(md.cloud_vertadr[cl])
        if n < 4:
            continue
        try:
            hull = ConvexHull(np.asarray(md.vert_pos[a:a + n], dtype=np.float64))
        except Exception:   # a flat cloud has no hull graph: support vertex only
            continue
        nb = [set() for _ in range(n)]
        for simp in hull.simplices:
            for i in simp:
                nb[int(i)].update(int(j) for j in simp if j != i)
        for v in range(n):
            adr[a + v], num[a + v] = len(adj), len(nb[v])
            adj.extend(sorted(nb[v]))
    out = (adr, num, np.asarray(adj if adj else [0], dtype=np.int32))
    _GRAPH_CACHE[key] = out
    return out


class MarshalledModel:
    """Owns the numpy buffers a GqModelDesc points into (keep alive for as long as the struct is in use)."""

    def __init__(self, md: ModelDesc, *, qpos0=None, feet_geom_names=None, terrain_limits=(1e4, -1e4, 1e4, -1e4),
                 timestep=None, solver=SOLVER_PGS, iterations=100, tolerance=1e-8, floor=None,
                 noise_floor=0.0, boxes=None, hfield=None, self_collision=None, mesh_graph=True):
        self.md = md
        self._keep = []
        d = GqModelDesc()
        d.struct_size = C.sizeof(GqModelDesc)
        nvert = int(md.vert_pos.shape[0])
        for k in ('nq', 'nv', 'nu', 'nbody', 'njnt', 'ngeom'):
            setattr(d, k, int(getattr(md, k)))
        d.ncloud, d.nvert = int(len(md.cloud_vertnum)), nvert
        d.timestep = float(md.timestep if timestep is None else timestep)
        d.gravity = (C.c_double * 3)(*md.gravity)
        d.cone, d.impratio, d.integrator = int(md.cone), float(md.impratio), int(md.integrator)
        q0 = np.array(md.qpos0 if qpos0 is None else qpos0, dtype=np.float64)
        boxes = list(boxes or [])
        from scipy.spatial.transform import Rotation as _Rot
        box_arrays = dict(
            box_pos=[b['pos'] for b in boxes], box_size=[b['size'] for b in boxes],
            box_mat=[_Rot.from_quat(np.asarray(b['quat'], float), scalar_first=True).as_matrix().ravel() for b in boxes],
            box_friction=[b['friction'] for b in boxes], box_margin=[b['margin'] for b in boxes], box_gap=[b['gap'] for b in boxes],
            box_solmix=[b['solmix'] for b in boxes], box_solref=[b['solref'] for b in boxes], box_solimp=[b['solimp'] for b in boxes],
            box_condim=[b['condim'] for b in boxes], box_priority=[b['priority'] for b in boxes])
        d.nbox = len(boxes)
        self.boxes = boxes
        from .selfcol import geom_capsules, self_pairs
        if self_collision is None:   # MuJoCo's behaviour (robot geoms collide with each other) wherever the solver supports it
            self_collision = int(solver) == SOLVER_NEWTON
        if self_collision not in (True, False, 'convex', 'capsule'):
            raise ValueError("self_collision must be True / 'convex' (MuJoCo's mesh-mesh routine), 'capsule' (capsule proxies for mesh / cylinder geoms) or False")
        d.self_convex = 0 if self_collision == 'capsule' else 1
        self.self_collision = 'off' if not self_collision else ('capsule' if self_collision == 'capsule' else 'convex')
        pairs = self_pairs(md) if self_collision else np.zeros((0, 2), np.int32)
        box_arrays.update(selfpair_geom1=pairs[:, 0], selfpair_geom2=pairs[:, 1], geom_capsule=geom_capsules(md))
        d.nselfpair = int(len(pairs))
        self.self_pairs = pairs
        for name, ctype in GqModelDesc._fields_:
            if ctype in (_I, _D) and not name.startswith('plane_') and not name.startswith('vert_adj') and name != 'support_grid':   # (the optional tables are filled below)
                src = q0 if name == 'qpos0' else (box_arrays[name] if name in box_arrays else getattr(md, name))
                arr = np.ascontiguousarray(src, dtype=np.int32 if ctype is _I else np.float64)
                if arr.size == 0:
                    arr = np.zeros(1, dtype=arr.dtype)
                self._keep.append(arr)
                setattr(d, name, arr.ctypes.data_as(ctype))
        fl = dict(friction=(1.0, 0.005, 0.0001), margin=0.0, gap=0.0, solmix=1.0, solref=(0.02, 1.0),
                  solimp=(0.9, 0.95, 0.001, 0.5, 2.0), condim=3, priority=0)
        fl.update(floor or {})
        d.floor_friction = (C.c_double * 3)(*fl['friction'])
        d.floor_margin, d.floor_gap, d.floor_solmix = fl['margin'], fl['gap'], fl['solmix']
        d.floor_solref = (C.c_double * 2)(*fl['solref'])
        d.floor_solimp = (C.c_double * 5)(*fl['solimp'])
        d.floor_condim, d.floor_priority = fl['condim'], fl['priority']
        names = feet_geom_names or {k: k for k in LEG_NAMES}
        d.feet_geomid = (C.c_int32 * 4)(*[md.geom_names.index(names[k]) for k in LEG_NAMES])
        d.terrain_limits = (C.c_double * 4)(*terrain_limits)
        d.meaninertia = float(md.meaninertia)
        kq = md.key_qpos[0] if len(md.key_qpos) else q0
        d.key_qpos = (C.c_double * 19)(*[float(v) for v in kq])
        d.solver, d.iterations, d.tolerance = int(solver), int(iterations), float(tolerance)
        d.noise_floor = float(noise_floor)
        # height field: dict(data=[nrow][ncol] in [0, 1], size=(rx, ry, elevation, base), pos=(x, y, z), + geom defaults)
        self.hfield = hfield
        if hfield is not None:
            data = np.ascontiguousarray(hfield['data'], dtype=np.float32)
            if data.ndim != 2:
                raise ValueError('hfield data must be a 2-D array [nrow][ncol]')
            self._keep.append(data)
            d.hfield_nrow, d.hfield_ncol = int(data.shape[0]), int(data.shape[1])
            d.hfield_data = data.ctypes.data_as(C.POINTER(C.c_float))
            d.hfield_size = (C.c_double * 4)(*[float(v) for v in hfield['size']])
            d.hfield_pos = (C.c_double * 3)(*[float(v) for v in hfield.get('pos', (0.0, 0.0, 0.0))])
            hg = dict(friction=(1.0, 0.005, 0.0001), margin=0.0, gap=0.0, solmix=1.0, solref=(0.02, 1.0),
                      solimp=(0.9, 0.95, 0.001, 0.5, 2.0), condim=3, priority=0)
            hg.update({k: v for k, v in hfield.items() if k in hg})
            d.hfield_friction = (C.c_double * 3)(*hg['friction'])
            d.hfield_margin, d.hfield_gap, d.hfield_solmix = hg['margin'], hg['gap'], hg['solmix']
            d.hfield_solref = (C.c_double * 2)(*hg['solref'])
            d.hfield_solimp = (C.c_double * 5)(*hg['solimp'])
            d.hfield_condim, d.hfield_priority = hg['condim'], hg['priority']
        # hull-versus-plane support tables (optional in the C-ABI: NULL = every chunk of a cloud is scanned)
        if nvert > 0 and any(int(c) > 64 for c in md.cloud_vertnum):
            pv, pm, po, pc = plane_support_tables(md)
            pv, pm, po, pc = np.ascontiguousarray(pv, dtype=np.float64), np.ascontiguousarray(pm, dtype=np.int32), np.ascontiguousarray(po, dtype=np.int32), np.ascontiguousarray(pc, dtype=np.float64)
            self._keep += [pv, pm, po, pc]
            d.plane_cap = pc.ctypes.data_as(_D)
            d.plane_grid = PLANE_GRID
            d.plane_vert_pos = pv.ctypes.data_as(_D)
            d.plane_mask = pm.ctypes.data_as(_I)
            d.plane_order = po.ctypes.data_as(_I)
        # support-function grids of the clouds (optional: NULL = every hull pair past the oriented boxes goes to the convex routine)
        if nvert > 0:
            sg = np.ascontiguousarray(support_grids(md), dtype=np.float64)
            self._keep.append(sg)
            d.support_grid = sg.ctypes.data_as(_D)
        # hull graphs of the mesh clouds (optional in the C-ABI: NULL = a mesh meets a plane at its support vertex only)
        if nvert > 0 and mesh_graph and hull_graphs_enabled(md):
            ga, gn, gl = hull_graphs(md)
            ga, gn, gl = np.ascontiguousarray(ga, dtype=np.int32), np.ascontiguousarray(gn, dtype=np.int32), np.ascontiguousarray(gl, dtype=np.int32)
            self._keep += [ga, gn, gl]
            d.nadj = int(gn.sum())
            d.vert_adjadr, d.vert_adjnum, d.vert_adj = ga.ctypes.data_as(_I), gn.ctypes.data_as(_I), gl.ctypes.data_as(_I)
        self.desc = d


def obs_ids_from_names(names):
    ids = []
    for n in names:
        if n not in OBS_NAMES:
            raise ValueError(f'Invalid observation name: {n}, available obs: {ALL_OBS}')
        ids.append(OBS_NAMES.index(n))
    return ids
