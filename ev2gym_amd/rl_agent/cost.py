"""cost_function plugins: the safety-RL split of a reward into (profit, cost).

Mirrors the names and call signature of the reference's `ev2gym/rl_agent/cost.py`.  A cost function is evaluated on the
host through the single-env facade (`ev2gym_amd.env.EV2Gym(cost_function=...)`) at the point where the reference
calls it (ev2gym_env.py:434-438), with the same arguments as the reward function, and its value is returned in
`info["cost"]` (terminal step: `env.cost`).
"""
import math

__all__ = ["transformer_overload_usrpenalty_cost", "ProfitMax_TrPenalty_UserIncentives_safety"]

_OVERLOAD_WEIGHT = 100.0          # per kW above / below a transformer's limits
_DISSATISFACTION_WEIGHT = 100.0   # times exp(-10 * satisfaction score) per departing EV


def _penalties(env, user_satisfaction_list):
    overload = sum(tr.get_how_overloaded() for tr in env.transformers)
    unhappy = sum(math.exp(-10 * score) for score in user_satisfaction_list)
    return _OVERLOAD_WEIGHT * overload + _DISSATISFACTION_WEIGHT * unhappy


def transformer_overload_usrpenalty_cost(env, total_costs, user_satisfaction_list, *args):
    """cost.py:8-18 -- exactly what ProfitMax_TrPenalty_UserIncentives subtracts from the profit, as a positive cost."""
    return _penalties(env, user_satisfaction_list)


def ProfitMax_TrPenalty_UserIncentives_safety(env, total_costs, user_satisfaction_list, *args):
    """cost.py:22-27 -- the profit alone; the penalties are reported by the cost function above."""
    return total_costs
