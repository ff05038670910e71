"""The names ETGRL's drivers import from the (absent) rlschool package, served by this repo, so that switching a driver is
an import swap:

    import rlschool                                                    ->  from paddlerobotics_amd import rlschool_names as rlschool
    from rlschool.quadrupedal.envs.utilities.ETG_model import ETG_layer, ETG_model
    from rlschool.quadrupedal.envs.env_wrappers.MonitorEnv import Param_Dict, Random_Param_Dict
    from rlschool.quadrupedal.envs.env_builder import SENSOR_MODE
    from rlschool.quadrupedal.robots import robot_config               ->  from paddlerobotics_amd.rlschool_names import (...)
    (train.py:19-27, pretrain.py:24, env_test.py:7)

`rlschool.make_env('Quadrupedal', ...)` keeps the reference's keywords (train.py:305-309); add num_envs / device for the
batched env, or single=True for one robot behind the numpy / scalar surface.  The values of Param_Dict are the reward
weights the drivers overwrite from their command line anyway (train.py:255-261, defaults :481-487); Random_Param_Dict and
SENSOR_MODE carry the keys the drivers set (train.py:253-254, :262-272)."""
import enum

from . import a1_model as _A
from .env import DEFAULT_SENSOR_MODE as _DEFAULT_SENSOR_MODE, make_env  # noqa: F401
from .etg import ETG_layer, etg_joint_action as _etg_joint_action  # noqa: F401

Param_Dict = {k: v for k, v in _A.DEFAULT_REWARD_PARAM.items() if k != "done"}          # torso feet up tau stand badfoot footcontact
Random_Param_Dict = {"random_dynamics": 0, "random_force": 0}
SENSOR_MODE = dict(_DEFAULT_SENSOR_MODE, ETG_obs=0, footpose=0, dynamic_vec=0, force_vec=0, noise=0)


class ETG_model:
    """A fixed (w, b) pair behind the layer: joint-space ETG action at time t (the drivers import the name; none calls it)."""

    def __init__(self, layer, w, b):
        self.layer, self.w, self.b = layer, w, b

    def forward(self, t):
        return _etg_joint_action(self.layer, self.w, self.b, t)


class robot_config:          # rlschool.quadrupedal.robots.robot_config (deployment/robots/robot_config.py:24-40)
    class MotorControlMode(enum.Enum):
        POSITION = 1
        TORQUE = 2
        HYBRID = 3
        PWM = 4
