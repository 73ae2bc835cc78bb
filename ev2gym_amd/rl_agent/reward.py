"""reward_function plugins (mirror of the reference's ev2gym/rl_agent/reward.py, same names and signature).

Each function has the reference signature `f(env, total_costs, user_satisfaction_list, *args) -> float` and works on
the single-env facade (`ev2gym_amd.env.EV2Gym`) exactly like the reference's works on its env.  Functions that
carry `_ev2g_kind` are additionally FUSED into the HIP step kernel: when one of them is passed as
`reward_function=` the engine computes the reward on the GPU and the Python body below is never called.
Any other callable is a user plugin: it is evaluated on the host through the facade (explicit slow path).
"""
import math


def SquaredTrackingErrorReward(env, *args):
    """reward.py:7-14"""
    t = env.current_step - 1
    return -(min(env.power_setpoints[t], env.charge_power_potential[t]) - env.current_power_usage[t]) ** 2


def ProfitMax_TrPenalty_UserIncentives(env, total_costs, user_satisfaction_list, *args):
    """reward.py:34-44"""
    reward = total_costs
    for tr in env.transformers:
        reward -= 100 * tr.get_how_overloaded()
    for score in user_satisfaction_list:
        reward -= 100 * math.exp(-10 * score)
    return reward


def profit_maximization(env, total_costs, user_satisfaction_list, *args):
    """reward.py:78-87"""
    reward = total_costs
    for score in user_satisfaction_list:
        reward -= 100 * math.exp(-10 * score)
    return reward


def SimpleReward(env, *args):
    """reward.py:60-65 """
    t = env.current_step - 1
    return -(env.power_setpoints[t] - env.current_power_usage[t]) ** 2


def _tracking_gap(env, *limits):
    """(min(setpoint, potential, *limits) - usage) at the step that was just simulated."""
    t = env.current_step - 1
    return min(env.power_setpoints[t], env.charge_power_potential[t], *limits) - env.current_power_usage[t]


def SqTrError_TrPenalty_UserIncentives(env, _, user_satisfaction_list, *args):
    """reward.py:16-32: tracking error capped by transformer 0's limit, overload and dissatisfaction penalties"""
    gap = _tracking_gap(env, env.transformers[0].max_power[env.current_step - 1])
    penalty = sum(100 * tr.get_how_overloaded() for tr in env.transformers)
    penalty += sum(1000 * (1 - score) for score in user_satisfaction_list)
    return -gap ** 2 - penalty


def SquaredTrackingErrorRewardWithPenalty(env, *args):
    """reward.py:46-58: an extra -100 when nothing was delivered although there was potential the step before"""
    t = env.current_step - 1
    idle = env.current_power_usage[t] == 0 and env.charge_power_potential[t - 1] != 0
    return -_tracking_gap(env) ** 2 - (100 if idle else 0)


def MinimizeTrackerSurplusWithChargeRewards(env, *args):
    """reward.py:67-76: quadratic penalty on exceeding the setpoint, linear bonus for delivered power"""
    t = env.current_step - 1
    usage, sp = env.current_power_usage[t], env.power_setpoints[t]
    surplus = usage - sp
    return (-(surplus ** 2) if sp < usage else 0) + usage


def V2G_costs_simple(env, total_costs, user_satisfaction_list, *args):
    """reward.py:151-154"""
    return total_costs


def V2G_profitmax(env, total_costs, user_satisfaction_list, *args):
    """reward.py:120-148: profit minus 100 per kWh that a departing EV is short of its desired capacity"""
    short = sum(100 * (ev.desired_capacity - ev.current_capacity) for ev in env.departing_evs
                if ev.desired_capacity > ev.current_capacity)
    return total_costs - short


def _v2_user_costs(env):
    """The user term of the *V2 rewards (reward.py:173-207): every connected EV that can no longer reach its desired capacity at
    full power pays 0.05 * (shortfall at this point of its stay)^2, every departing EV 0.05 * (final shortfall)^2."""
    user_costs = 0
    for cs in env.charging_stations:
        for ev in cs.evs_connected:
            if ev is not None:
                min_steps_to_full = (ev.desired_capacity - ev.current_capacity) / (ev.max_ac_charge_power / (60 / env.timescale))
                departing_step = ev.time_of_departure - env.current_step
                if min_steps_to_full > departing_step:
                    min_capacity_at_time = ev.desired_capacity - ((departing_step + 1) * ev.max_ac_charge_power / (60 / env.timescale))
                    user_costs += -(0.05 * (min_capacity_at_time - ev.current_capacity) ** 2)
    for ev in env.departing_evs:
        if ev.desired_capacity > ev.current_capacity:
            user_costs += -0.05 * (ev.desired_capacity - ev.current_capacity) ** 2
    return user_costs


def V2G_profitmaxV2(env, total_costs, user_satisfaction_list, *args):
    """reward.py:156-211"""
    return total_costs + _v2_user_costs(env)


def pst_V2G_profitmaxV2(env, total_costs, user_satisfaction_list, *args):
    """reward.py:278-339: V2G_profitmaxV2 plus 1000 x the (negative) excess of the power over the setpoint"""
    t = env.current_step - 1
    pst_violation = 0
    if env.power_setpoints[t] < env.current_power_usage[t]:
        pst_violation += env.power_setpoints[t] - env.current_power_usage[t]
    return total_costs + _v2_user_costs(env) + 1000 * pst_violation


ProfitMax_TrPenalty_UserIncentives._ev2g_kind = 0
SquaredTrackingErrorReward._ev2g_kind = 1
profit_maximization._ev2g_kind = 2
SqTrError_TrPenalty_UserIncentives._ev2g_kind = 3
SquaredTrackingErrorRewardWithPenalty._ev2g_kind = 4
SimpleReward._ev2g_kind = 5
MinimizeTrackerSurplusWithChargeRewards._ev2g_kind = 6
V2G_costs_simple._ev2g_kind = 7
V2G_profitmax._ev2g_kind = 8
V2G_profitmaxV2._ev2g_kind = 9
pst_V2G_profitmaxV2._ev2g_kind = 10
