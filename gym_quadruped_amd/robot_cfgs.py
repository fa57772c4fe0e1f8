"""Robot registry - host-side mirror of the reference's ``gym_quadruped/robot_cfgs.py`` (:7-60).

Same names, same fields, same matching rules (substring match for mini_cheetah / hyqreal1 / hyqreal2 / spot,
exact match for go1 / go2 / aliengo / b2 / pegasus; anything else - including a bare ``'hyqreal'`` - raises
``ValueError``).  ``mjcf_filename`` keeps the reference's relative path; its stem selects the compiled table in
``model_data/``.
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import Iterable, Optional

import numpy as np


def _default_feet():
    return {'FL': 'FL', 'FR': 'FR', 'RL': 'RL', 'RR': 'RR'}


def _default_leg_joints():
    return {leg: [f'{leg}_hip_joint', f'{leg}_thigh_joint', f'{leg}_calf_joint'] for leg in ('FL', 'FR', 'RL', 'RR')}


@dataclass
class RobotConfig:
    """Configuration of a quadruped robot model."""

    mjcf_filename: str
    hip_height: float  # height of the hip joint in the nominal stand pose
    qpos0_js: Optional[Iterable] = None  # joint-space zero configuration override
    feet_geom_names: dict = field(default_factory=_default_feet)
    leg_joints: dict = field(default_factory=_default_leg_joints)
    accel_name: Optional[str] = None
    gyro_name: Optional[str] = None
    imu_site_name: Optional[str] = None


_EXACT = {
    'go1': ('go1/go1.xml', 0.3),
    'go2': ('go2/go2.xml', 0.28),
    'aliengo': ('aliengo/aliengo.xml', 0.35),
    'b2': ('b2/b2.xml', 0.485),
    'pegasus': ('pegasus/pegasus.xml', 0.5),
}
_SUBSTR = [('hyqreal1', 'hyqreal1/hyqreal1.xml', 0.498), ('hyqreal2', 'hyqreal2/hyqreal2.xml', 0.498),
           ('spot', 'spot/spot.xml', 0.46)]


def get_robot_config(robot_name: str) -> RobotConfig:
    """Name -> :class:`RobotConfig`, with the reference's precedence (robot_cfgs.py:35-58)."""
    name = robot_name.lower()
    if 'mini_cheetah' in name:
        return RobotConfig(mjcf_filename='mini_cheetah/mini_cheetah.xml', hip_height=0.225,
                           qpos0_js=[0, -np.pi / 2, 0] * 2 + [0, np.pi / 2, 0] * 2)
    for key in ('go1', 'go2', 'aliengo', 'b2'):
        if name == key:
            return RobotConfig(mjcf_filename=_EXACT[key][0], hip_height=_EXACT[key][1])
    for sub, fname, hh in _SUBSTR:
        if sub in name:
            return RobotConfig(mjcf_filename=fname, hip_height=hh)
    if name == 'pegasus':
        return RobotConfig(mjcf_filename=_EXACT['pegasus'][0], hip_height=_EXACT['pegasus'][1])
    raise ValueError(f'Unknown robot name: {robot_name}')
