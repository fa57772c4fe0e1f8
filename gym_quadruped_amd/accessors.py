"""Batched counterparts of the reference's on-demand getters (quadruped_env.py:488-1016) - the "MPC accessors".

The reference computes these from ``mjData`` whenever the user asks; right after ``step()`` that is exactly what
``_get_obs`` stored, so here they are *views*:

* getters whose value is one of the ``ALL_OBS`` observables (base velocities / errors / acceleration in both frames,
  feet positions and velocities, contact state and ground reaction forces, base configuration, Euler angles, heading
  frame, gravity vector, kinetic energy, work) read the observation row the step kernel assembled.  Either list the
  observable in ``state_obs_names`` or construct the env with ``accessors=True`` (all of ``ALL_OBS`` is then appended
  to the row the kernel writes; the user-visible observation dict is unchanged).
* getters that read MuJoCo internals of the last forward pass - ``mj_fullM`` (``legs_mass_matrix``, ``get_base_inertia``),
  ``qfrc_bias`` / ``qfrc_passive``, ``body(i).xpos`` (``hip_positions``), ``subtree_com`` (``com``), ``mj_jac``
  (``feet_jacobians``), ``mjData.contact`` / ``mj_contactForce`` (``contacts``, ``mj_contactForce``) - read two extra rows the
  PRODUCTION step kernel writes per env when the env was built with ``accessors=True`` (``gq_batch_set_outputs``: the
  tree-sparse inertia, bias forces, body poses, foot points; the contact list with its forces).  Without
  ``accessors=True`` they fall back to the kernel's inspection record (instrumented variant, ~18 % slower, one 8.4 KB
  record per env-step; the first use needs one ``step()`` / ``reset()`` afterwards to fill it).

Like the reference values they describe the LAST forward pass: positions / Jacobians / M of the pose before the
integration step, velocities of the new state (quirk B3).  Everything has a leading env axis and stays on the GPU.
"""
from __future__ import annotations

import ctypes as C

import numpy as np
import torch

from . import _lib
from .utils.quadruped_utils import LegsAttr

LEGS = ('FL', 'FR', 'RL', 'RR')   # canonical kernel order


class _DevPtr:
    """Minimal __cuda_array_interface__ carrier so torch can wrap library-owned device memory without a copy."""

    def __init__(self, ptr, shape, typestr='<f4'):
        self.__cuda_array_interface__ = {'shape': tuple(shape), 'typestr': typestr, 'data': (int(ptr), False), 'version': 2}


class AccessorsMixin:
    # ------------------------------------------------------------------ plumbing
    def _acc(self, name):
        v = self._obs_views.get(name)
        if v is None:
            v = self._extra_views.get(name)
        if v is None:
            raise _lib.GqError(f"'{name}' is not assembled by this env: add it to state_obs_names or pass accessors=True")
        return v

    def _per_leg(self, flat, width):
        """[N, 4*width] ordered by legs_order -> LegsAttr of [N, width] views."""
        out = {}
        for k, leg in enumerate(self.legs_order):
            out[leg] = flat[:, k * width:(k + 1) * width]
        return LegsAttr(**out)

    @staticmethod
    def _frame(frame):
        if frame not in ('world', 'base'):
            raise ValueError(f"Invalid frame: {frame} != 'world' or 'base'")
        return '' if frame == 'world' else ':base'

    _DYN_FIELDS = {'qfrc_bias': ('BIAS', 18), 'xpos': ('XPOS', 39), 'xmat': ('XMAT', 117), 'foot_pos': ('FOOT', 12)}

    def _record(self, field):
        """[N, count] view of a by-product of the last forward pass: a slice of the production kernel's dynamics row
        (accessors=True), else of the inspection record (instrumented kernel, all envs)."""
        if getattr(self, '_dyn', None) is not None:
            from .cabi import GQ_DYN
            if field == 'M':
                return self._dense_mass()
            key, cnt = self._DYN_FIELDS[field]
            return self._dyn[:, GQ_DYN[key]:GQ_DYN[key] + cnt]
        self._ensure_rec_tensor()
        if self._rec_filled_at is None:
            raise _lib.GqError('the inspection record is empty: call step() or reset() once after the first use of a '
                               'dynamics accessor (legs_mass_matrix, legs_qfrc_bias, feet_jacobians, hip_positions, com, ...)')
        off, cnt = C.c_int32(), C.c_int32()
        _lib.check(self._L.gq_debug_field(field.encode(), C.byref(off), C.byref(cnt)), 'gq_debug_field')
        return self._rec_tensor[:, off.value:off.value + cnt.value]

    def _ensure_rec_tensor(self):
        """Switch the inspection record on for every env (instrumented kernel variant from the next launch on) and map it."""
        if getattr(self, '_rec_tensor', None) is None:
            self.enable_debug(self.num_envs)
            ptr, n, stride = C.c_void_p(), C.c_int32(), C.c_int32()
            _lib.check(self._L.gq_debug_device_buffer(self._hbatch, C.byref(ptr), C.byref(n), C.byref(stride)), 'gq_debug_device_buffer')
            self._rec_tensor = torch.as_tensor(_DevPtr(ptr.value, (n.value, stride.value)), device=self.device)
            self._rec_filled_at = None

    def _dense_mass(self):
        """mj_fullM: [N, 324] dense joint-space inertia out of the dynamics row's tree-sparse storage (leg dof 6 + j keeps its
        base columns and the columns of its own leg up to itself; the base block is full)."""
        from .cabi import GQ_DYN
        if getattr(self, '_mass_index', None) is None:
            src = np.full((18, 18), GQ_DYN['STRIDE'], dtype=np.int64)          # default: a zero column appended below
            for a in range(6):
                for b in range(6):
                    src[a, b] = GQ_DYN['MB'] + 6 * a + b
            for j in range(12):
                leg, dep = divmod(j, 3)
                for k in range(6):
                    src[6 + j, k] = src[k, 6 + j] = GQ_DYN['MC'] + 9 * j + k
                for c in range(dep + 1):
                    src[6 + j, 6 + 3 * leg + c] = src[6 + 3 * leg + c, 6 + j] = GQ_DYN['MC'] + 9 * j + 6 + c
            self._mass_index = torch.as_tensor(src.ravel(), device=self.device)
        padded = torch.cat([self._dyn, torch.zeros(self.num_envs, 1, dtype=torch.float32, device=self.device)], dim=1)
        return padded.index_select(1, self._mass_index)

    def contacts(self):
        """mjData.contact of the last forward pass as tensors (needs accessors=True): dict with 'ncon' [N] and, per contact
        slot k < 12 (MuJoCo's order; slots >= ncon are zero), 'geom1' (-1: a world geom), 'geom2', 'dist', 'pos' (x / y relative
        to the pre-step base x / y), 'frame' [N, 12, 3, 3] (rows: normal, tangent 1, tangent 2), 'dim', 'force' [N, 12, 6]
        (mj_contactForce: contact-frame force and torque), 'mu'."""
        if getattr(self, '_contacts', None) is None:
            raise _lib.GqError('contacts() needs an env built with accessors=True')
        from .cabi import GQ_CON_MAX, GQ_CON_REC
        rows = self._contacts
        rec = rows[:, 8:].reshape(self.num_envs, GQ_CON_MAX, GQ_CON_REC)
        return dict(ncon=rows[:, 0].to(torch.int32), nefc=rows[:, 1].to(torch.int32), geom1=rec[:, :, 0].to(torch.int32), geom2=rec[:, :, 1].to(torch.int32),
                    dist=rec[:, :, 2], pos=rec[:, :, 3:6], frame=rec[:, :, 6:15].reshape(self.num_envs, GQ_CON_MAX, 3, 3), dim=rec[:, :, 15].to(torch.int32),
                    force=rec[:, :, 16:22], mu=rec[:, :, 22])

    def mj_contactForce(self, contact_id: int):
        """``mujoco.mj_contactForce(m, d, id, result)`` (reference :852) for contact ``contact_id`` of every env: [N, 6] (zeros
        where the env has fewer contacts).  ``gq_contact_force`` on the contact rows (accessors=True)."""
        if getattr(self, '_contacts', None) is None:
            raise _lib.GqError('mj_contactForce needs an env built with accessors=True')
        out = torch.empty(self.num_envs, 6, dtype=torch.float32, device=self.device)
        stream = torch.cuda.current_stream(self.device).cuda_stream
        _lib.check(self._L.gq_contact_force(self._hbatch, int(contact_id), out.data_ptr(), stream), 'gq_contact_force')
        return out

    def _note_step(self):
        if getattr(self, '_rec_tensor', None) is not None:
            self._rec_filled_at = int(self._launches)

    # ------------------------------------------------------------------ the MuJoCo entry points of the lower boundary, batched
    def mj_jac(self, point, body, qpos=None):
        """``mujoco.mj_jac(m, d, jacp, jacr, point, body)`` (reference :728-735) for every env at pose ``qpos`` (default: the
        current state): ``point`` [N, 3] world coordinates, ``body`` MuJoCo body id or name.  Returns (jacp, jacr), each
        [N, 3, 18] float32.  One production kernel launch (``gq_jac``), no inspection record."""
        if isinstance(body, str):
            body = self.mjModel.body_names.index(body)
        N, dev = self.num_envs, self.device
        q = self._qpos if qpos is None else torch.as_tensor(qpos, dtype=torch.float64, device=dev).reshape(-1, 19).expand(N, 19).contiguous()
        p = torch.as_tensor(point, dtype=torch.float64, device=dev).reshape(-1, 3).expand(N, 3).contiguous()
        jp = torch.empty(N, 3, 18, dtype=torch.float32, device=dev); jr = torch.empty_like(jp)
        stream = torch.cuda.current_stream(dev).cuda_stream
        _lib.check(self._L.gq_jac(self._hbatch, q.data_ptr(), int(body), p.data_ptr(), jp.data_ptr(), jr.data_ptr(), stream), 'gq_jac')
        return jp, jr

    def mj_ray(self, pnt, vec, return_geom=False):
        """``mujoco.mj_ray`` against the static geoms of the scene (floor, world boxes, height field; reference
        sensors/heightmap.py:90-99 uses flg_static=1): ``pnt`` [N, R, 3] world origins, ``vec`` [N, R, 3] directions.
        Returns distances [N, R] in units of |vec| (-1: no hit) and, if asked, the geom hit (0 floor, 1 + box, 1 + nbox
        height field, -1 none)."""
        N, dev = self.num_envs, self.device
        o = torch.as_tensor(pnt, dtype=torch.float64, device=dev).reshape(N, -1, 3).contiguous()
        d = torch.as_tensor(vec, dtype=torch.float32, device=dev).reshape(N, -1, 3).contiguous()
        R = o.shape[1]
        dist = torch.empty(N, R, dtype=torch.float32, device=dev)
        geom = torch.empty(N, R, dtype=torch.int32, device=dev) if return_geom else None
        stream = torch.cuda.current_stream(dev).cuda_stream
        _lib.check(self._L.gq_ray(self._hbatch, o.data_ptr(), d.data_ptr(), int(R), dist.data_ptr(), None if geom is None else geom.data_ptr(), stream), 'gq_ray')
        return (dist, geom) if return_geom else dist

    def mj_forward(self, ctrl=None, stage=0):
        """``mujoco.mj_forward`` (stage 0, reference :1321) / ``mujoco.mj_step1`` (stage 1, :376, :384) for every env without
        advancing the state; the results are read like the reference reads mjData afterwards - through the dynamics
        accessors (legs_mass_matrix, legs_qfrc_bias, feet_jacobians, hip_positions, com, ...) or ``debug_internals``.  Uses
        the instrumented kernel variant for all envs (~20 % slower than step's production kernel)."""
        # gq_forward writes the inspection record (and, on an accessors=True env, the dynamics / contact rows of the forward
        # pose): the record must be on whatever backs the getters - not through _record(), which answers from the
        # dynamics row as soon as there is one and never switches the record on
        self._ensure_rec_tensor()
        c = None if ctrl is None else torch.as_tensor(ctrl, dtype=torch.float32, device=self.device).reshape(-1, self.mjModel.nu).expand(self.num_envs, -1).contiguous()
        stream = torch.cuda.current_stream(self.device).cuda_stream
        _lib.check(self._L.gq_forward(self._hbatch, int(stage), None if c is None else c.data_ptr(), self._st, self._out, stream), 'gq_forward')
        self._launches += 1
        self._note_step()

    def mj_step1(self):
        self.mj_forward(stage=1)

    # ------------------------------------------------------------------ observation-backed getters
    def base_lin_vel(self, frame='world'):
        return self._acc('base_lin_vel' + self._frame(frame))

    def base_ang_vel(self, frame='world'):
        return self._acc('base_ang_vel' + self._frame(frame))

    def base_lin_vel_err(self, frame='world'):
        return self._acc('base_lin_vel_err' + self._frame(frame))

    def base_ang_vel_err(self, frame='world'):
        return self._acc('base_ang_vel_err' + self._frame(frame))

    def base_lin_acc(self, frame='world'):
        return self._acc('base_lin_acc' + self._frame(frame))

    def feet_pos(self, frame='world') -> LegsAttr:
        return self._per_leg(self._acc('feet_pos' + self._frame(frame)), 3)

    def feet_vel(self, frame: str = 'world', relative: bool = False) -> LegsAttr:
        return self._per_leg(self._acc(('feet_vel_rel' if relative else 'feet_vel') + self._frame(frame)), 3)

    def feet_contact_state(self, frame='world', ground_reaction_forces=False):
        """(contact_state, ground reaction forces) per leg.  The reference also returns the per-foot lists of MjContact
        objects (:836-855); a batch has no such objects - per-contact detail is in the inspection record."""
        # 'contact_state' is ALWAYS in FL, FR, RL, RR order, whatever legs_order says (the reference builds it with
        # to_list() without `order=`, quadruped_env.py:1194-1195, quirk B5); the forces below follow legs_order
        cs_row = self._acc('contact_state')
        cs = LegsAttr(**{leg: cs_row[:, k] > 0.5 for k, leg in enumerate(LEGS)})
        if not ground_reaction_forces:
            return cs, None
        return cs, None, self._per_leg(self._acc('contact_forces' + self._frame(frame)), 3)

    @property
    def base_configuration(self):
        """[N, 4, 4] homogeneous base pose X_B (reference :961-970)."""
        so3, pos = self.heading_free_rotation, self._acc('base_pos')
        X = torch.zeros(self.num_envs, 4, 4, dtype=torch.float32, device=self.device)
        X[:, :3, :3] = so3; X[:, :3, 3] = pos; X[:, 3, 3] = 1.0
        return X

    @property
    def heading_free_rotation(self):
        """[N, 3, 3] base rotation matrix (observable 'base_ori_SO3', row-major)."""
        return self._acc('base_ori_SO3').reshape(self.num_envs, 3, 3)

    @property
    def base_ori_euler_xyz(self):
        return self._acc('base_ori_euler_xyz')

    @property
    def heading_orientation_SO3(self):
        """[N, 3, 3] rotation about z by the base yaw (reference :989-997)."""
        yaw = self._acc('base_ori_euler_xyz')[:, 2]
        c, s = torch.cos(yaw), torch.sin(yaw)
        R = torch.zeros(self.num_envs, 3, 3, dtype=torch.float32, device=self.device)
        R[:, 0, 0] = c; R[:, 0, 1] = -s; R[:, 1, 0] = s; R[:, 1, 1] = c; R[:, 2, 2] = 1.0
        return R

    @property
    def gravity_vector(self):
        return self._acc('gravity_vector:base')

    @property
    def kinetic_energy(self):
        return self._acc('kinetic_energy')[:, 0]

    @property
    def work(self):
        return self._acc('work')[:, 0]

    # ------------------------------------------------------------------ inspection-record-backed getters
    def _full_mass_matrix(self):
        return self._record('M').reshape(self.num_envs, 18, 18)

    @property
    def legs_mass_matrix(self) -> LegsAttr:
        M = self._full_mass_matrix()
        out = {}
        for leg in LEGS:
            idx = torch.as_tensor(self.legs_qvel_idx[leg], device=self.device)
            out[leg] = M[:, idx][:, :, idx]
        return LegsAttr(**out)

    def get_base_inertia(self):
        return self._full_mass_matrix()[:, 3:6, 3:6]

    @property
    def legs_qfrc_bias(self) -> LegsAttr:
        b = self._record('qfrc_bias')
        return LegsAttr(**{leg: b[:, self.legs_qvel_idx[leg]] for leg in LEGS})

    @property
    def legs_qfrc_passive(self) -> LegsAttr:
        """qfrc_passive of these models is joint damping only: -damping * qvel (mj_passive)."""
        damp = torch.as_tensor(np.asarray(self.mjModel.dof_damping, dtype=np.float32), device=self.device)
        p = -damp * self._qvel
        return LegsAttr(**{leg: p[:, self.legs_qvel_idx[leg]] for leg in LEGS})

    def _body_poses(self):
        """World positions [N, 13, 3] / rotations [N, 13, 3, 3] of the robot bodies at the last forward pass.  The record
        keeps positions relative to the base x/y of that pass (fp32 never sees the 10 km spawn offsets): it is added back
        from the stored base position observable when available, else the record is returned base-relative in x/y."""
        xpos = self._record('xpos').reshape(self.num_envs, 13, 3).clone()
        xmat = self._record('xmat').reshape(self.num_envs, 13, 3, 3)
        return xpos, xmat

    def hip_positions(self, frame='world') -> LegsAttr:
        """Hip body origins (reference :564-595).  World frame = relative to the pre-step base x/y plus that offset."""
        xpos, xmat = self._body_poses()
        names = list(self.mjModel.body_names)
        out = {}
        for leg in LEGS:
            b = names.index(f'{leg}_hip') - 1   # record bodies exclude the world body
            p = xpos[:, b] + self._base_xy_offset()
            out[leg] = p if frame == 'world' else torch.einsum('nij,ni->nj', xmat[:, 0], p)  # reference: R.T @ xpos (quirk: no translation)
        if frame not in ('world', 'base'):
            raise ValueError(f"Invalid frame: {frame} != 'world' or 'base'")
        return LegsAttr(**out)

    def _base_xy_offset(self):
        """[N, 3] pre-step base x/y (z = 0): new qpos minus h * new qvel (the Euler update of the free joint translation)."""
        off = torch.zeros(self.num_envs, 3, dtype=torch.float32, device=self.device)
        off[:, :2] = (self._qpos[:, :2] - float(self._sim_dt) * self._qvel[:, :2].double()).float()
        return off

    @property
    def com(self):
        """The reference's `com` (:918-929): sum_i body_mass[i] * subtree_com[i] / total_mass (sic - subtree COMs weighted by
        body masses), evaluated on the poses of the last forward pass."""
        xpos, xmat = self._body_poses()
        md = self.mjModel
        mass = torch.as_tensor(np.asarray(md.body_mass[1:], dtype=np.float32), device=self.device)          # 13 robot bodies
        ipos = torch.as_tensor(np.asarray(md.body_ipos[1:], dtype=np.float32).reshape(13, 3), device=self.device)
        xipos = xpos + torch.einsum('nbij,bj->nbi', xmat, ipos)
        parent = [int(p) - 1 for p in md.body_parentid[1:]]
        sub_m = mass.clone().repeat(self.num_envs, 1)
        sub_mx = xipos * mass[None, :, None]
        for b in range(12, 0, -1):           # children have larger ids than parents
            sub_m[:, parent[b]] = sub_m[:, parent[b]] + sub_m[:, b]
            sub_mx[:, parent[b]] = sub_mx[:, parent[b]] + sub_mx[:, b]
        subtree_com = sub_mx / sub_m[:, :, None]
        # world body (id 0): its subtree is the whole robot, its own mass 0 -> contributes nothing to the weighted sum
        com = (subtree_com * mass[None, :, None]).sum(1) / mass.sum()
        return com + self._base_xy_offset()

    def feet_jacobians(self, frame: str = 'world', return_rot_jac: bool = False):
        """mj_jac of the foot geom centre on its calf body (:681-740): [N, 3, 18] per leg (and the rotational one)."""
        if frame not in ('world', 'base'):
            raise ValueError(f"Invalid frame: {frame} != 'world' or 'base'")
        xpos, xmat = self._body_poses()
        md, N, dev = self.mjModel, self.num_envs, self.device
        foot = self._record('foot_pos').reshape(N, 4, 3)           # canonical FL FR RL RR, base-x/y relative like xpos
        jpos = torch.as_tensor(np.asarray(md.jnt_pos, dtype=np.float32).reshape(-1, 3)[1:], device=dev)    # 12 hinges
        jax = torch.as_tensor(np.asarray(md.jnt_axis, dtype=np.float32).reshape(-1, 3)[1:], device=dev)
        anchor = xpos[:, 1:] + torch.einsum('nbij,bj->nbi', xmat[:, 1:], jpos)   # hinge j lives on body j + 1
        axis = torch.einsum('nbij,bj->nbi', xmat[:, 1:], jax)
        Rb = xmat[:, 0]
        jp, jr = {}, {}
        for k, leg in enumerate(LEGS):
            p = foot[:, k]
            Jp = torch.zeros(N, 3, 18, dtype=torch.float32, device=dev)
            Jr = torch.zeros(N, 3, 18, dtype=torch.float32, device=dev)
            Jp[:, 0, 0] = 1.0; Jp[:, 1, 1] = 1.0; Jp[:, 2, 2] = 1.0
            r = p - xpos[:, 0]
            for a in range(3):   # free joint rotation: body-frame axes
                ax = Rb[:, :, a]
                Jp[:, :, 3 + a] = torch.cross(ax, r, dim=1)
                Jr[:, :, 3 + a] = ax
            for d in self.legs_qvel_idx[leg]:
                j = d - 6
                Jp[:, :, d] = torch.cross(axis[:, j], p - anchor[:, j], dim=1)
                Jr[:, :, d] = axis[:, j]
            if frame == 'base':
                Jp = torch.einsum('nji,njk->nik', Rb, Jp); Jr = torch.einsum('nji,njk->nik', Rb, Jr)
            jp[leg], jr[leg] = Jp, Jr
        return (LegsAttr(**jp), LegsAttr(**jr)) if return_rot_jac else LegsAttr(**jp)

    def feet_jacobians_dot(self, frame: str = 'world', return_rot_jac: bool = False):
        """mj_jacDot of the foot points (:742-797): time derivative of ``feet_jacobians`` along the current velocity, the point
        moving with its calf body; poses of the last forward pass, velocities of the current state (like every getter
        here).  For a hinge with world axis a on a body of angular velocity w, anchored at c:  d/dt [a x (p - c)] =
        (w x a) x (p - c) + a x (v_p - v_c);  the free joint's rotational columns use the base axes the same way."""
        if frame not in ('world', 'base'):
            raise ValueError(f"Invalid frame: {frame} != 'world' or 'base'")
        xpos, xmat = self._body_poses()
        md, N, dev = self.mjModel, self.num_envs, self.device
        qv = self._qvel
        foot = self._record('foot_pos').reshape(N, 4, 3)
        jpos = torch.as_tensor(np.asarray(md.jnt_pos, dtype=np.float32).reshape(-1, 3)[1:], device=dev)
        jax = torch.as_tensor(np.asarray(md.jnt_axis, dtype=np.float32).reshape(-1, 3)[1:], device=dev)
        anchor = xpos[:, 1:] + torch.einsum('nbij,bj->nbi', xmat[:, 1:], jpos)
        axis = torch.einsum('nbij,bj->nbi', xmat[:, 1:], jax)
        Rb, xb = xmat[:, 0], xpos[:, 0]
        w_base = torch.einsum('nij,nj->ni', Rb, qv[:, 3:6])
        v_base = qv[:, 0:3]
        cross = lambda a, b: torch.cross(a, b, dim=1)
        jp, jr = {}, {}
        for k, leg in enumerate(LEGS):
            p = foot[:, k]
            dofs = self.legs_qvel_idx[leg]                       # hip, thigh, calf dofs of this leg
            # angular velocity of each chain body and velocity of a point r carried by chain body i (i = 0 hip .. 2 calf)
            w = [w_base]
            for d in dofs:
                w.append(w[-1] + axis[:, d - 6] * qv[:, d:d + 1])

            def vel(r, upto):   # velocity of world point r attached to the body after `upto` hinges of the leg
                v = v_base + cross(w_base, r - xb)
                for d in dofs[:upto]:
                    v = v + cross(axis[:, d - 6], r - anchor[:, d - 6]) * qv[:, d:d + 1]
                return v
            vp = vel(p, 3)
            Jp = torch.zeros(N, 3, 18, dtype=torch.float32, device=dev)
            Jr = torch.zeros(N, 3, 18, dtype=torch.float32, device=dev)
            for a in range(3):
                e = Rb[:, :, a]
                ed = cross(w_base, e)
                Jp[:, :, 3 + a] = cross(ed, p - xb) + cross(e, vp - v_base)
                Jr[:, :, 3 + a] = ed
            for i, d in enumerate(dofs):
                a_d, c_d = axis[:, d - 6], anchor[:, d - 6]
                ad = cross(w[i], a_d)                            # the axis turns with the body the hinge hangs from
                Jp[:, :, d] = cross(ad, p - c_d) + cross(a_d, vp - vel(c_d, i))
                Jr[:, :, d] = ad
            if frame == 'base':
                Jp = torch.einsum('nji,njk->nik', Rb, Jp); Jr = torch.einsum('nji,njk->nik', Rb, Jr)
            jp[leg], jr[leg] = Jp, Jr
        return (LegsAttr(**jp), LegsAttr(**jr)) if return_rot_jac else LegsAttr(**jp)
