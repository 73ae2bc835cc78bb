"""cost_function plugins (mirror of the reference's ev2gym/rl_agent/cost.py; evaluated on the host through the
single-env facade, `EV2Gym(cost_function=...)`, exactly where the reference calls them: ev2gym_env.py:434-438)."""
import math


def transformer_overload_usrpenalty_cost(env, total_costs, user_satisfaction_list, *args):
    """cost.py:8-18: what ProfitMax_TrPenalty_UserIncentives subtracts from the profit, as a positive cost"""
    return (sum(100 * tr.get_how_overloaded() for tr in env.transformers)
            + sum(100 * math.exp(-10 * score) for score in user_satisfaction_list))


def ProfitMax_TrPenalty_UserIncentives_safety(env, total_costs, user_satisfaction_list, *args):
    """cost.py:22-27: the profit alone (the penalties live in the cost above)"""
    return total_costs
