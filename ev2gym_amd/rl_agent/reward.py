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
    """reward.py:60-65 (host-evaluated plugin: not fused)"""
    t = env.current_step - 1
    return -(env.power_setpoints[t] - env.current_power_usage[t]) ** 2


ProfitMax_TrPenalty_UserIncentives._ev2g_kind = 0
SquaredTrackingErrorReward._ev2g_kind = 1
profit_maximization._ev2g_kind = 2
