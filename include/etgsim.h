/*
 * etgsim.h -- C-ABI of the MI355X-native batched A1 quadruped simulator.
 *
 * This is the drop-in boundary of the hot path: the Gym surface
 *     env.reset(ETG_w=, ETG_b=, ...) -> (obs, info)
 *     env.step(action, donef=)       -> (obs, reward, done, info)
 * that PaddleRobotics' QuadrupedalRobots/ETGRL drives
 * (reference: ETGRL/train.py:131,147,186,195,215,228; pretrain.py:131,138;
 *  model/Dynamic_parallel_model.py:55,61; env_test.py:47,53).
 * In the reference that surface is served by `rlschool.make_env('Quadrupedal')`
 * (train.py:305-309) on top of one pybullet client per robot; here it is served
 * by hand-written gfx950 kernels for N robots at once.  The Python host
 * (paddlerobotics_amd/env.py) binds these entry points with ctypes and passes
 * raw device pointers (torch tensor .data_ptr()).
 *
 * Conventions
 *   - every function returns 0 on success, a negative ETG_ERR_* code otherwise,
 *     and never throws across the ABI; etg_last_error() returns a message.
 *   - the caller owns obs/action/reward/done/info device buffers; the library
 *     owns its internal SoA state.  All work is enqueued on the passed HIP
 *     stream (a hipStream_t passed as void*); no hidden synchronisation.
 *   - a handle is bound to one device; calls on one handle are serialised by
 *     the caller.
 *   - there is NO CPU fallback in this library: with no HIP device every entry
 *     point that would launch work fails with ETG_ERR_NO_DEVICE.
 */
#ifndef ETGSIM_H_
#define ETGSIM_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ETG_NUM_LEGS 4
#define ETG_NUM_MOTORS 12
#define ETG_ACT_DIM 12
#define ETG_HYBRID_DIM 60 /* HYBRID motor commands: 12 x (q_des, kp, qd_des, kd, tau_ff), laikago_motor.py:30-37 */
#define ETG_OBS_DIM 49   /* train.py:311 with the default sensor set */
#define ETG_STATE_DIM 37 /* pos3 quat4(xyzw) linvel3 angvel3 (world) q12 qd12      */
#define ETG_DYN_DIM 48   /* flattened dynamic_param dict, train.py:112-126          */
#define ETG_RBF_H 20     /* ETG_H, train.py:479                                     */
#define ETG_INFO_DIM 64  /* per-env info row, layout below                          */

/* info row layout (floats) -- the keys callers read */
#define ETG_INFO_TORSO 0
#define ETG_INFO_FEET 1
#define ETG_INFO_UP 2
#define ETG_INFO_TAU 3
#define ETG_INFO_STAND 4
#define ETG_INFO_BADFOOT 5
#define ETG_INFO_FOOTCONTACT 6
#define ETG_INFO_DONE 7
#define ETG_INFO_VELX 8
#define ETG_INFO_ETG_ACT 9       /* 12: info["ETG_act"], env_test.py:54          */
#define ETG_INFO_JOINT_ANGLE 21  /* 12: info["joint_angle"], Dynamic_parallel_model.py:63 */
#define ETG_INFO_OBS_IMU 33      /* 6 : info["obs-IMU"] = rpy,drpy (un-normalised) */
#define ETG_INFO_FOOT_CONTACT 39 /* 4 : info["FootContactSensor"], EnvWrapper.py:108 */
#define ETG_INFO_REAL_ACTION 43  /* 12: info["real_action"] = q_des, EnvWrapper.py:108 */
#define ETG_INFO_BASE 55         /* 3 : base position (world)                      */
#define ETG_INFO_RPY 58          /* 3 : true roll pitch yaw                        */
#define ETG_INFO_ENERGY 61       /* 1 : sum_ticks sum_j |tau_j qd_j| dt            */
#define ETG_INFO_STEPS 62        /* 1 : control steps since reset                  */
#define ETG_INFO_SWEEPS 63       /* 1 : contact-solver sweeps EXECUTED for this robot during the step, summed over its ticks:
                                        the count of the slowest robot sharing its wavefront (what the step cost) */

/* optional sensors of train.py:268-271 (sensor_mode ETG_obs / footpose / dynamic_vec / force_vec): columns of the
 * [N, ETG_EXTRA_DIM] row etg_extra_sensors() writes.  rlschool's own definitions are absent from the reference
 * tree; these are this library's.                                                        */
#define ETG_EXTRA_ETG_OBS 0      /* 20: RBF activations r(t) of the ETG layer at the time of the observation */
#define ETG_EXTRA_FOOTPOSE 20    /* 12: foot positions in the base frame from the OBSERVED motor angles
                                        (a1.py:113-140 foot_positions_in_base_frame)                         */
#define ETG_EXTRA_DYNAMIC 32     /* 48: the robot's dynamic_param row mapped back to the [-1,1] box of
                                        param2dynamic_dict (train.py:112-126)                                 */
#define ETG_EXTRA_FORCE 80       /* 3 : external force on the trunk, world frame, newtons (set + random push) */
#define ETG_EXTRA_DIM 84

enum {
  ETG_OK = 0,
  ETG_ERR_BAD_ARG = -1,
  ETG_ERR_NO_DEVICE = -2,
  ETG_ERR_HIP = -3,
  ETG_ERR_ALLOC = -4,
  ETG_ERR_STATE = -5
};

/* Rigid-body inertial parameters of one link, in the link frame. */
typedef struct EtgLink {
  double mass;
  double com[3];
  double inertia[6]; /* about the COM: xx yy zz xy xz yz */
} EtgLink;

/* The A1 model.  Values come from paddlerobotics_amd/a1_model.py; the kinematic
 * ones are the reference's (deployment/robots/a1.py:52,70-73,83,98-100), the
 * inertial ones are recalled from pybullet_data/a1/a1.urdf (absent from the
 * reference tree -- SURVEY App. B). Base frame = trunk COM frame.            */
typedef struct EtgRobotModel {
  EtgLink trunk;             /* com must be 0 (base frame is the trunk COM frame) */
  EtgLink hip[4], thigh[4], calf[4], foot[4]; /* per leg FR,FL,RR,RL (already mirrored) */
  double hip_origin[4][3];   /* hip joint origin in the base frame = a1.py HIP_OFFSETS */
  double thigh_y[4];         /* +-0.08505, sign (-1)^(leg+1), a1.py:100          */
  double upper_len;          /* 0.2, a1.py:98 */
  double lower_len;          /* 0.2, a1.py:99 */
  double foot_radius;        /* 0.02 collision sphere                           */
  double init_pos[3];        /* 0,0,0.32  a1.py:52 */
  double pose_ori[12];       /* [0,.9,-1.8]*4  a1.py:83 */
  double base_foot[12];      /* ETG nominal foot positions in the base frame    */
  double etg_mean[12];       /* EnvWrapper.py:50-53 */
  double etg_std[12];        /* EnvWrapper.py:54-55 */
} EtgRobotModel;

typedef struct EtgConfig {
  int32_t num_envs;
  int32_t action_repeat;   /* physics ticks per control step (13)              */
  int32_t settle_ticks;    /* ticks holding pose_ori on reset (500, a1.py:294) */
  int32_t solver_iters;    /* PGS sweeps per tick: the count, or the cap when solver_residual > 0 */
  int32_t enable_action_interp; /* minitaur.py:1384-1401                        */
  int32_t enable_action_filter; /* action_filter.py (Butterworth), default off */
  int32_t obs_normal;      /* `normal` kwarg, EnvWrapper.py:66-87              */
  int32_t terrain;         /* 0 = flat plane z=0, 1 = heightfield              */
  double sim_dt;           /* 0.002                                            */
  double erp;              /* Baumgarte factor for penetrating contacts (0.2)  */
  double contact_margin;   /* speculative contact distance (0.02)             */
  double warmstart;        /* warm-start factor of the normal impulses (pybullet's server: 0.1; Bullet's own default 0.85) */
  double torque_limit;     /* <=0: none (A1 passes none, a1.py:256-274)        */
  double etg_T, etg_T2, etg_amp, etg_sigma_sq, etg_phase[2]; /* train.py:296-297 */
  double etg_dt;           /* control period 0.026                             */
  double reward_w[8];      /* torso feet up tau stand badfoot footcontact done */
  double reward_p;         /* train.py:472 */
  double vel_d;            /* train.py:470 */
  double filter_b[3], filter_a[3]; /* Butterworth coefficients (host computes) */
  /* heightfield (terrain==1): row-major [hf_ny][hf_nx] heights, cell size, origin */
  int32_t hf_nx, hf_ny;
  double hf_cell, hf_x0, hf_y0;
  /* kernel mapping: 16 = one robot per 16-lane DPP row (fills the chip at 4096 robots), 4 = one
   * robot per quad (one leg per lane; 4x the robots per wave, for bigger batches), 0 = auto
   * (16 if num_envs <= 4096 -- <= 8192 with body_contacts 1 / 2 -- else 4). Same results to fp32 roundoff, same state layout. */
  int32_t lanes_per_robot;
  /* terrain variants: the heightfield's hf_ny rows are hf_bands equal bands stacked along y; robot e
   * walks on band e % hf_bands (its own y axis, clamped inside the band). 0 or 1 = one shared terrain.
   * This is how per-episode stair/slope parameters (train.py:48-50) coexist in one batch.          */
  int32_t hf_bands;
  /* motor control mode (robot_config.MotorControlMode, train.py mode_map): 0 = POSITION -- the action is a
   * joint-angle residual on pose_ori + ETG and the PD law of laikago_motor.py:165-173 makes the torque;
   * 1 = TORQUE -- the action IS the 12 motor torques (laikago_motor.py:140-143), no ETG/pose added;
   * 2 = HYBRID -- the action row has ETG_HYBRID_DIM = 60 floats, per motor (q_des, kp, qd_des, kd, tau_ff), and
   * tau = -kp (q - q_des) - kd (qd - qd_des) + tau_ff (laikago_motor.py:152-167), no ETG/pose added.
   * The reset settle always runs under POSITION control like a1.py:289-304.                          */
  int32_t motor_mode;
  /* A1._ClipMotorCommands (deployment/robots/a1.py:439-457, MAX_MOTOR_ANGLE_CHANGE_PER_STEP = 0.2): when
   * > 0, every sub-step's position command is clipped to GetMotorAngles() +- this many radians -- the motor angle as the
   * control observation sees it: delayed by the robot's control latency (the blend of minitaur.py:1172-1193) and wrapped to
   * [-pi, pi], without the sensor noise the reference adds to every reading.  0 = off (the default of the reference's
   * constructor).                                                                                      */
  double clip_motor_commands;
  /* body contacts (Bullet collides every link's URDF shape with the ground, a1.py:276-287; here spheres of
   * knee_radius stand in for the link shapes, besides the four foot spheres).  Served by both lane mappings, flat ground and
   * heightfield.  default_config / make_env switch mode 2 ON (the reference's robot cannot pass its shins through the floor). */
  int32_t body_contacts;   /* 0 off (toe spheres only); 1 one contact per leg on a sphere at the knee (the calf joint origin,
                            * carried by the thigh); 2 one contact per leg on the DEEPEST of three spheres: knee, shin midpoint
                            * (carried by the calf), trunk corner next to the leg's hip (trunk_half below).  A contact of modes
                            * 1 / 2 has a normal row and two friction rows (body_friction) like a foot's, solved after the feet's
                            * rows of the same kind.  Its normal row is warm-started like a foot's (`warmstart` x its impulse
                            * of the tick before; friction rows from 0) when the contact is one persistent point -- mode 1,
                            * mode 2 with body_blend > 0 -- as Bullet's persistent manifold does (ABI version 2); under the
                            * hard deepest-of-three choice (body_blend = 0) it starts cold.  3 all three spheres of every leg at once, a FRICTIONLESS
                            * normal row each (24 contact rows per robot; served by the 4-lanes-per-robot mapping, which the
                            * setting selects; the fused closed-loop call is not available with it)                      */
  double knee_radius;
  /* `ETG` kwarg of make_env (train.py:305-309, Dynamic_parallel_model.py:49 runs with ETG=0): 0 switches the
   * trajectory generator off -- the position command is pose_ori + action, info["ETG_act"] and the ETG
   * observation columns are zero.  1 (the default of default_config) = pose_ori + ETG(t) + action.       */
  int32_t enable_etg;
  /* joint-limit stops (a1.py:186-195 UPPER_BOUND / LOWER_BOUND through the URDF limits Bullet enforces): when
   * != 0 a joint that has left [joint_lower, joint_upper] and still moves outward is stopped inelastically
   * (DESIGN.md section 2); 0 = no limits.  default_config / make_env switch them ON (Bullet always enforces the URDF's;
   * the reference's own recorded gait reaches the calf joint's upper bound).                                */
  int32_t joint_limits;
  double joint_lower[3], joint_upper[3];   /* hip, thigh, calf (rad) */
  double trunk_half[3];                    /* half extents of the trunk's collision box (body_contacts = 2), base frame */
  /* Stopping rule of the contact solve.  > 0: after every projected Gauss-Seidel sweep of a tick the robot's residual
   *     max over its active rows r of ((lambda_r - lambda_r at the start of the sweep) * A_rr)^2
   * (A_rr = the row's diagonal Delassus entry: the square of the velocity change the row's own impulse change produced) is
   * compared with solver_residual; the tick stops sweeping when it is <= solver_residual or after solver_iters sweeps,
   * whichever comes first (at least one sweep).  This is the least-squares-residual exit of Bullet's sequential-impulse
   * loop (stepSimulation, minitaur.py:244): pybullet documents numSolverIterations = 50 and
   * solverResidualThreshold = 1e-7, the defaults of default_config.  0: exactly solver_iters sweeps every tick.      */
  double solver_residual;
  /* friction cone handling of a foot's two tangent rows: 0 = after both rows the pair is projected on the disc of radius
   * mu lambda_n (isotropic Coulomb cone); 1 = each direction clamped on its own to +-mu lambda_n inside its row solve
   * (the friction pyramid of a sequential-impulse solver with two friction directions).                              */
  int32_t friction_model;
  /* pd_latency of the robot class (minitaur.py:100,130-132,1195-1199): the motor model's PD law reads the joint angles and
   * velocities this many seconds old -- blended from the two history readings that bracket the latency, exactly like the
   * control-latency observation (minitaur.py:1172-1193) -- instead of the current ones.  0 (the reference's default, A1
   * passes none) = the true state.  Applies to every sub-step, the reset settle included.  The delayed damping term is
   * physics, not a defect: with the default gains the loop is marginal from ~2.5 ms (two fp32 evaluations settle 1e-3 rad
   * apart) and unstable from ~4 ms (the settle ends in non-finite numbers and the robot reports done).                 */
  double pd_latency;
  /* Warm start of a foot's two friction rows: their impulses of the previous tick times this factor start the solve.  0 (the
   * default) = Bullet's multibody solver, which warm-starts the normal row only (with `warmstart`) and restarts friction rows
   * from zero.                                                                                                        */
  double warmstart_friction;
  /* Added to a contact's distance before the velocity target is formed (Bullet: penetration = distance + m_linearSlop;
   * pybullet's physics server sets the slop to 1e-5 m, the default of default_config).                                */
  double contact_slop;
  /* Combined coefficient of restitution of foot and ground (SetFootRestitution, minitaur.py:1112-1122; Bullet multiplies the
   * two bodies' coefficients, so with the reference's ground plane at its default 0 the product is 0 = the default here):
   * when a foot approaches the ground faster than Bullet's restitution velocity threshold (0.2 m/s) at the start of a tick,
   * its normal row's velocity target is raised by restitution x the approach speed.                                   */
  double foot_restitution;
  /* Friction coefficient of the body contacts of body_contacts 1 / 2 (their two friction rows are solved like a foot's, on the
   * disc body_friction x the contact's normal impulse; friction_model 1: the pyramid).  Bullet: the product of the two bodies'
   * lateralFriction -- 0.5 for a URDF link without a <contact> element (btCollisionObject's default; the reference only ever
   * changes the FEET's, minitaur.py:1100-1110) x the ground's 1.0 = 0.5, the default of default_config.  0 = frictionless. */
  double body_friction;
  /* body_contacts 2: softness (m) of the choice among the leg's three spheres.  0 = the DEEPEST one is the contact (rounds 5's
   * model).  A shin lying along the ground has its knee and shin-midpoint spheres at the same depth: the hard choice then hops
   * between two points 0.1 m apart from tick to tick, decided by the last bit of the arithmetic -- no two implementations agree on
   * such a robot, and its ticks need the most sweeps.  > 0: the contact's impulse is DISTRIBUTED over the spheres with weights
   * w_i ~ exp(-(d_i - d_min) / body_blend) (d_i: the sphere's distance to the ground): point, normal, distance and Jacobian rows
   * are the weighted means.  A sphere deeper than the others by a few body_blend carries everything -- the hard choice -- and two
   * equally deep ones share the load, as the two ends of a line contact do.  default_config: 1e-3.                          */
  double body_blend;
} EtgConfig;

typedef struct EtgHandle EtgHandle;

/* ---- lifecycle ---------------------------------------------------------- */
int etg_create(const EtgConfig* cfg, const EtgRobotModel* model, int device, EtgHandle** out);
void etg_destroy(EtgHandle* h);
const char* etg_last_error(void);
/* ABI version: bumped whenever EtgConfig / EtgRobotModel change layout or an entry point changes meaning (2: round 6 --
 * EtgConfig gained body_friction in round 5 without a bump; finished episodes are no longer simulated by the fused rollouts).
 * etg_config_size() / etg_model_size(): sizeof(EtgConfig) / sizeof(EtgRobotModel) as the LIBRARY was compiled -- a binding
 * compares them with its own layout before the first etg_create (paddlerobotics_amd/_lib.py does). */
int etg_version(void);
int etg_config_size(void);
int etg_model_size(void);
int etg_lanes_per_robot(const EtgHandle* h); /* the mapping etg_create resolved (4 or 16) */

/* ---- parameters (device pointers, float32) ------------------------------
 * dyn   : [N,48] physical-unit dynamic_param rows (layout of train.py:112-126:
 *         latency_ms, footfriction, basemass, baseinertia3, legmass3,
 *         leginertia12, kp12, kd12, gravity3) or NULL to keep.
 * etg_w : [N,3,20] if per_env else [3,20]; etg_b : [N,3] / [3]; NULL keeps.
 * mask  : [N] uint8 or NULL (= all): only masked envs are updated.           */
int etg_set_params(EtgHandle* h, const float* dyn, const float* etg_w, const float* etg_b,
                   int per_env, const uint8_t* mask, void* stream);
/* heightfield heights [hf_ny*hf_nx] float32 device pointer (terrain==1)      */
int etg_set_heightfield(EtgHandle* h, const float* heights, void* stream);

/* external force on the trunk COM, world frame, newtons: force [N,3] float32 device pointer, applied on
 * every tick of the following steps until replaced; NULL clears it (random_force of train.py:254,
 * README "Add external random force").                                                              */
int etg_set_external_force(EtgHandle* h, const float* force, void* stream);

/* motor strength ratios (Minitaur.SetMotorStrengthRatios, minitaur.py:1280-1294 -> LaikagoMotorModel.set_strength_ratios,
 * laikago_motor.py:67-76): ratios [N,12] float32 device pointer, one factor per motor on the motor model's output torque
 * (laikago_motor.py:138,167: before the torque clip; in TORQUE mode on the commanded torque); NULL = 1 for everybody.
 * mask [N] bytes or NULL = all.  The reset settle runs under the motor model, so the touched robots settle again at their
 * next reset.                                                                                                        */
int etg_set_motor_strength(EtgHandle* h, const float* ratios, const uint8_t* mask, void* stream);

/* random pushes, sampled on the device (random_param['random_force'], train.py:254; rlschool's own schedule is
 * absent, this one is the repo's): call once per control step before etg_step. A robot without an active push
 * starts one with probability `prob`: a horizontal force of magnitude U(fmin, fmax) N in a uniform direction,
 * held for `duration_steps` control steps, then removed. Counter-based RNG keyed by (seed, robot, call index),
 * so a run is reproducible. Robots being reset should have their push cleared with etg_clear_pushes.  */
int etg_random_pushes(EtgHandle* h, uint64_t seed, float prob, int duration_steps, float fmin, float fmax,
                      void* stream);
int etg_clear_pushes(EtgHandle* h, const uint8_t* mask, void* stream);

/* Gaussian sensor noise (minitaur.py:1206-1211 _AddSensorNoise; stdev[5] in the order of SENSOR_NOISE_STDDEV,
 * minitaur.py:102: motor angle, motor velocity, motor torque, base rpy, base rpy rate). Added to the delayed
 * readings that make up the observation (joint angles before MapToMinusPiToPi, joint velocities, rpy, rpy rate;
 * torques are not part of the 49-float row); rewards, termination and the motor model read the true state, as in
 * the reference. Counter-based RNG keyed by (seed, robot, observation index, channel): reproducible, identical
 * through etg_step and the fused rollouts. NULL or all-zero stdev switches it off (the default).            */
int etg_set_sensor_noise(EtgHandle* h, const float* stdev, uint64_t seed);

/* start offsets xy [N,2] (m, added to init_pos x / y) used by the following resets of the masked robots
 * (mask NULL = all; xy NULL = zero) -- the `x_noise` of env.reset (train.py:131,186,215; BCtrain.py:89: a start
 * position jitter so the stairs are not always met in the same gait phase; rlschool's distribution is absent, the
 * caller draws it). The robot settles AT the offset position; on flat ground, where the settle is translation
 * invariant, a cached settle is shifted instead of re-simulated.                                            */
int etg_set_reset_offsets(EtgHandle* h, const float* xy, const uint8_t* mask, void* stream);

/* ---- the hot path ------------------------------------------------------- */
/* reset masked envs (NULL = all): place at init pose, settle, write obs[N,49]
 * rows of the reset envs (other rows untouched).                             */
int etg_reset(EtgHandle* h, const uint8_t* mask, float* obs, void* stream);
/* one control step for all envs. action [N,12] (already scaled by act_bound,
 * train.py:147; [N,60] in the HYBRID motor mode); donef [N] uint8 or NULL; outputs obs [N,49], reward [N],
 * done [N] uint8, info [N,64] or NULL.                                       */
int etg_step(EtgHandle* h, const float* action, const uint8_t* donef, float* obs,
             float* reward, uint8_t* done, float* info, void* stream);
/* etg_step for the sub-batch [env0, env0 + count) only (ABI version 2).  Every pointer is the WHOLE batch's array, indexed by
 * the robot's number exactly as in etg_step; rows and states of the other robots are not touched.  env0 and count are
 * multiples of 16 robots (the last range may end at N).  Robots are independent, so G ranges enqueued on G streams are the
 * same computation as one etg_step, robot by robot and bit for bit -- but each sub-batch's next step can start when ITS slowest
 * wavefront has finished (the Gym loop of train.py:129-178 with one barrier per group instead of one per batch).  The
 * sensor-noise stream position moves on with the range that starts at robot 0: call the ranges of one control step in
 * ascending order.                                                                                                      */
int etg_step_range(EtgHandle* h, int env0, int count, const float* action, const uint8_t* donef, float* obs,
                   float* reward, uint8_t* done, float* info, void* stream);
/* etg_step followed by the reset of every robot whose `done` byte the step set (Gym-style auto-reset, on the device,
 * no host synchronisation): obs rows of those robots hold their reset observation, reward / done / info rows the
 * finished step.  While every robot has a cached settle (after a full etg_reset, until dynamic parameters, terrain or
 * heightfield start offsets change -- and again once masked etg_resets have settled every robot such a change touched)
 * step and restart are ONE launch; otherwise etg_step + etg_reset(mask = done).                                        */
int etg_step_autoreset(EtgHandle* h, const float* action, const uint8_t* donef, float* obs,
                       float* reward, uint8_t* done, float* info, void* stream);
/* Dynamics randomisation per EPISODE without a settle per control step (random_param["random_dynamics"], train.py:253: a new
 * draw of the 48 dynamic parameters at every reset).  A reset with new parameters needs a simulated settle (settle_ticks
 * ticks: tens of control steps' worth of time, whatever the number of robots), and with thousands of robots some episode
 * ends on nearly every step.  etg_prepare_next_dynamics derives the rows dyn [N,48] (device; rows of robots outside `mask`
 * are ignored, NULL = all) for the robots' NEXT episodes and runs their settle NOW, in one launch, on scratch state -- the
 * robots keep running on their current parameters.  The settled state replaces the robot's settle cache; the rows are
 * installed when the robot's episode ends inside etg_step_autoreset (same launch as the step, as before) or at its next
 * etg_reset.  A robot whose episode ends again before the next call starts over with the same rows.  Robots in the first 64
 * physics ticks of their episode are left out of the call (they still read their pre-reset history from the settle cache the
 * call replaces): their pending flag stays 0 and the caller's next refresh covers them.  Needs a full etg_reset
 * before (every robot with a cached settle); etg_set_params with new dynamics for a robot drops its pending rows.
 * etg_next_dynamics_pending copies the pending flags (1 = rows still waiting) to pending [N] bytes (device): the robots with 0
 * are the ones to draw new rows for.                                                                                   */
int etg_prepare_next_dynamics(EtgHandle* h, const float* dyn, const uint8_t* mask, void* stream);
int etg_next_dynamics_pending(EtgHandle* h, uint8_t* pending, void* stream);
/* per-robot episode statistics since the robot's last reset: return (sum of
 * rewards) and length (steps), both frozen after the first `done` (alive
 * masking; the batched counterpart of train.py:213-249 / pretrain.py:129-154).
 * Every etg_step updates them on device. ret [N] f32 / len [N] i32, or NULL.    */
int etg_episode_stats(EtgHandle* h, float* ret, int32_t* len, void* stream);
/* ---- fused rollouts: etg_rollout_openloop / _actions / _policy / _policy_record ------------------------------------------
 * AN ENDED EPISODE IS NOT SIMULATED ANY MORE (the default; ABI version 2).  The reference's episode loops leave at `done`
 * (pretrain.py:137-153, train.py:226-247); so does every fused rollout, per robot: once a robot's step has reported done
 * (termination; the accumulators' alive flag, cleared by etg_reset only), later steps of the same and of later rollout calls
 * do not touch it -- its state (etg_get_state) stays the terminal state, its row of `obs` the observation of the step that
 * ended the episode, return and length what they were.  Per-step outputs of the recording entry points read reward 0 and
 * done 1 for such steps; their other rows (observations, actions, joint angles, IMU) are NOT written -- the caller's buffer
 * keeps its content there.  etg_step / etg_step_autoreset (the Gym surface) step every robot, as before.
 * etg_set_rollout_mode(h, 1) brings back the behaviour of ABI version 1 -- finished robots are simulated on, only their
 * accumulators are masked -- for callers that measure it or look at a robot after its fall; 0 = the default.          */
int etg_set_rollout_mode(EtgHandle* h, int simulate_finished);
/* Measurement: the shader-clock cycles every wavefront spent in each launch of the LAST etg_rollout_openloop call, cycles
 * [launches][waves] int64 (device pointer; NULL = only report the two sizes).  A launch lasts as long as its slowest wave: the
 * sum over the launches of the largest entry against the mean of the waves' totals says how unevenly the contact solver's
 * sweeps load the wavefronts (bench.py: "imbalance").  A rollout is cut into launches of 400 control steps; the environment
 * variable ETG_ROLLOUT_CHUNK changes that for a process (a measurement aid: profiles/r06_chunk_sweep.txt).            */
int etg_rollout_wave_cycles(EtgHandle* h, int64_t* cycles, int capacity, int* launches, int* waves, void* stream);
/* open-loop rollout: n_steps x etg_step(action = 0) enqueued back-to-back
 * (pretrain.py:129-154), then etg_episode_stats. obs [N,49] (or NULL)
 * receives the final observation of every robot (see above: of a finished
 * robot, its last one).                                                       */
int etg_rollout_openloop(EtgHandle* h, int n_steps, float* obs, float* ret, int32_t* len,
                         void* stream);

/* n_steps control steps over a caller-supplied action tape, fused (<= 400 control steps per launch, robot state and control
 * variables in registers in between): actions [n_steps,N,12] (device, already scaled: what etg_step takes), step s applies
 * row block s.  The loops it serves know their commands in advance: the dynamics-identification evaluator replays 2 x 100
 * recorded joint targets and reads info["joint_angle"] and info["obs-IMU"] after every step
 * (model/Dynamic_parallel_model.py:53-77), the BC teacher replays (BCtrain.py:87-131).  Optional per-step outputs (NULL = not
 * recorded): rec_joint_angle [n_steps,N,12], rec_imu [n_steps,N,6] (rpy - first rpy, body rates: the info columns),
 * rec_obs [n_steps,N,49] (without sensor noise), rec_reward [n_steps,N], rec_done [n_steps,N] bytes.  obs [N,49] receives the
 * final observation; ret / len as etg_episode_stats (may be NULL).  The same source as n_steps calls of etg_step compiled
 * into another kernel: equal to rounding noise (multiply-add contraction differs between the two), not bit for bit.   */
int etg_rollout_actions(EtgHandle* h, const float* actions, int n_steps, float* obs, float* rec_joint_angle, float* rec_imu,
                        float* rec_obs, float* rec_reward, uint8_t* rec_done, float* ret, int32_t* len, void* stream);

/* ---- kinematics and optional sensors --------------------------------------- */
/* leg FK + analytic Jacobian of n joint-angle rows q [n,12] (device pointers), through the SAME device routine the
 * contact kinematics of the physics tick uses (leg_geometry): foot [n,12] = foot_positions_in_base_frame
 * (a1.py:131-140: hip-frame foot position + HIP_OFFSETS), jac [n,4,3,3] (may be NULL) = analytical_leg_jacobian
 * (a1.py:143-173), d foot / d (hip, thigh, calf angle) per leg.                                              */
int etg_leg_kinematics(EtgHandle* h, const float* q, int n, float* foot, float* jac, void* stream);
/* optional sensors (ETG_EXTRA_* columns) for the observation rows obs [N,49] the last etg_reset / etg_step /
 * rollout wrote: out [N, ETG_EXTRA_DIM].                                                                     */
int etg_extra_sensors(EtgHandle* h, const float* obs, float* out, void* stream);

/* ---- state access for parity tests (device pointers, [N,37] f32) --------- */
int etg_get_state(EtgHandle* h, float* state, void* stream);
int etg_set_state(EtgHandle* h, const float* state, void* stream);
/* the contact impulses of the last tick that the solver warm-starts from (EtgConfig.warmstart), [N,16] = per leg (foot n, t1, t2,
 * body contact's normal impulse).  etg_set_state zeroes them; a parity test that re-creates a robot's step from its state
 * installs them afterwards, so that the re-created step is the robot's own (tests/test_gpu_parity5.py).  The body contact's
 * friction rows, the legacy body_contacts = 3 rows and the joint stops are not warm-started (their entries stay 0). */
int etg_get_contact_impulses(EtgHandle* h, float* lam, void* stream);
int etg_set_contact_impulses(EtgHandle* h, const float* lam, void* stream);

/* ---- policy (the one dense contraction, model/mujoco_model.py:44-60) -----
 * act = tanh(W3 relu(W2 relu(W1 obs + b1) + b2) + b3) * act_scale
 * weights are torch [out,in] row-major fp32 device pointers; obs [N,in_dim],
 * act [N,12].  precision: 0 = fp32 MFMA (exact f32), 1 = bf16 MFMA.           */
typedef struct EtgPolicy EtgPolicy;
int etg_policy_create(int in_dim, int hidden, int out_dim, int device, EtgPolicy** out);
int etg_policy_load(EtgPolicy* p, const float* w1, const float* b1, const float* w2,
                    const float* b2, const float* w3, const float* b3, void* stream);
int etg_policy_forward(EtgPolicy* p, const float* obs, int n, float act_scale, int precision,
                       float* act, void* stream);
/* stochastic head (SAC.sample, alg/sac.py:65-76; used by the training rollouts of train.py:141):
 *   x = mean + exp(clamp(W_std h + b_std, -20, 2)) * noise,  act = tanh(x) * act_scale,
 *   logp (optional, [N]) = sum_j [log N(x_j; mean_j, std_j) - log(1 - tanh(x_j)^2 + 1e-6)]
 * noise [N,out_dim] is the caller's N(0,1) draw (device pointer), so the result is reproducible.    */
int etg_policy_load_std(EtgPolicy* p, const float* w_std, const float* b_std, void* stream);
int etg_policy_sample(EtgPolicy* p, const float* obs, int n, const float* noise, float act_scale,
                      int precision, float* act, float* logp, void* stream);
void etg_policy_destroy(EtgPolicy* p);

/* closed-loop rollout with a fixed actor in ONE kernel per 400 control steps (run_EStrain_episode / run_evaluate_episodes,
 * train.py:182-249): per step action = tanh(mean(obs)) * act_scale as etg_policy_forward computes it, then one
 * control step; a workgroup keeps its 16 robots' observations, actions and states on chip between the steps.
 * obs [N,49]: in = the current observation (as left by etg_reset / etg_step), out = the final one.  The actor sees the
 * columns [obs_col0, obs_col0 + in_dim) of it (0 and 49 for the teacher, 3 and 46 for the student of BCtrain.py:53-59).
 * ret / len as etg_episode_stats (may be NULL).  Whole workgroups only: num_envs % 16 == 0 on the 16-lanes-per-robot mapping
 * (16 robots per workgroup), num_envs % 64 == 0 on the 4-lane one (64 robots per workgroup, two stacked policy tiles per weight fetch). */
int etg_rollout_policy(EtgHandle* h, EtgPolicy* policy, int n_steps, float act_scale, int precision, int obs_col0,
                       float* obs, float* ret, int32_t* len, void* stream);

/* etg_rollout_policy that also RECORDS the episode for the replay memory of the ES-SAC loop (run_EStrain_episode with
 * es_rpm, train.py:213-249): per control step s and robot i, rec_obs [n_steps,N,49] = the observation the actor acted on,
 * rec_act [n_steps,N,12] = tanh(mean) (unscaled, as the reference stores it), rec_reward [n_steps,N], rec_done [n_steps,N]
 * (bytes).  next_obs of step s is rec_obs of step s + 1, of the last step the final obs.  Robots keep stepping after their
 * episode ended: the caller masks rows after a robot's first done.  noise [n_steps,N,12] (may be NULL): the caller's
 * N(0,1) draws for the STOCHASTIC actor of run_train_episode (agent.sample, train.py:143-144; alg/sac.py:65-76): the
 * action becomes tanh(mean + exp(clamp(log_std, -20, 2)) * noise) (needs etg_policy_load_std).                        */
int etg_rollout_policy_record(EtgHandle* h, EtgPolicy* policy, int n_steps, float act_scale, int precision, int obs_col0,
                              float* obs, const float* noise, float* rec_obs, float* rec_act, float* rec_reward,
                              uint8_t* rec_done, float* ret, int32_t* len, void* stream);

/* ---- ETG parameterisation (the step right before reset) ----
 * Batched Opt_with_points / LS_sol (train.py:59-110): for every candidate fit the
 * x- and z-row of the ETG weights through its 6 control points by LS_sol's
 * gradient descent (same stopping rule), anchored at w0 with weight lamb.
 * points [nb,6,2], feats [6,20] (ETG basis at the 6 control times), w0 [2,20]
 * (x row, z row); float64 device pointers. out_w [nb,3,20], out_b [nb,3] float64.  */
int etg_fit_etg(const double* points, int nb, const double* feats, const double* w0, double b0x,
                double b0z, double precision, double alpha, double lamb, int max_iter,
                double* out_w, double* out_b, void* stream);

/* ---- transitions for the off-policy learner ---------------------------------- */
/* The reference's loops append every transition to a replay memory: rpm.append(obs, action, reward, next_obs, terminal),
 * terminal = 1 - done (train.py:148-149,159,240-241), and sum the reward terms of `info` (train.py:150-156).  Batched
 * and masked: only robots whose episode is still running store a row.  The memory is five caller-owned device arrays of
 * max_size rows: mem_obs / mem_next_obs [max_size, obs_dim], mem_act [max_size, act_dim], mem_reward / mem_terminal
 * [max_size]; pos_count [2] (device, int64) = next slot to write, transitions ever stored.
 *
 * etg_replay_begin (BEFORE the step overwrites the observation buffer): slot[i] = ring slot of robot i's row, -1 for a
 * robot that is not alive (alive [n] bytes, NULL = all): its row is not stored; pos_count advanced, obs and act rows stored.  n must not exceed max_size.  act_scaled [n, act_dim]
 * (may be NULL) receives act_scale * act: the command of the step (action * act_bound, train.py:147).               */
int etg_replay_begin(const uint8_t* alive, int n, long long max_size, long long* pos_count, int32_t* slot,
                     const float* obs, int obs_dim, const float* act, int act_dim, float* mem_obs, float* mem_act,
                     float act_scale, float* act_scaled, void* stream);
/* etg_replay_end (after the step): reward, next_obs and terminal = 1 - done stored at the same slots.  If alive is
 * given: for alive robots the first n_sum columns of info [n, info_dim] are added to info_sum [n, n_sum + 1] and its last
 * column counts info[velx_col] >= 0.3 (train.py:156; velx_col < 0: off); then alive &= !done.  info / info_sum may be NULL. */
int etg_replay_end(const int32_t* slot, int n, const float* reward, const uint8_t* done, const float* next_obs, int obs_dim,
                   float* mem_reward, float* mem_terminal, float* mem_next_obs, const float* info, int info_dim,
                   int n_sum, int velx_col, float* info_sum, uint8_t* alive, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* ETGSIM_H_ */
