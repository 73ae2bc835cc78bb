"""Rule-based agents with the reference's `get_action(env)` surface (ev2gym/baselines/heuristics.py).

ChargeAsFastAsPossible / RandomAgent / DoNothing work on the single-env facade (returning `np.ndarray[number_of_ports]` like the reference) and on
`EV2GymVec` (returning a `[num_envs, number_of_ports]` array / device tensor).
"""
import math

import numpy as np


class ChargeAsFastAsPossible:
    """heuristics.py:152-166: every port at full charging power."""
    algo_name = "Charge As Fast As Possible"

    def __init__(self, verbose=False, **kwargs):
        self.verbose = verbose

    def get_action(self, env):
        if hasattr(env, "num_envs"):
            return env.full_like_actions(1.0)
        return np.ones(env.number_of_ports)


class RandomAgent:
    """heuristics.py:546-558: uniform samples of the action box."""
    algo_name = "Random Actions"

    def __init__(self, env=None, seed=0, **kwargs):
        self.rng = np.random.default_rng(seed)
        self._calls = 0
        self.seed = seed

    def get_action(self, env):
        low = -1.0 if env.v2g_enabled else 0.0
        if hasattr(env, "num_envs"):
            self._calls += 1
            return env.uniform_actions(self.seed * 1000003 + self._calls, low, 1.0)
        return self.rng.uniform(low, 1.0, env.number_of_ports)


class DoNothing:
    """heuristics.py:533-544: no port charges or discharges."""
    algo_name = "DO NOTHING"

    def __init__(self, verbose=False, **kwargs):
        self.verbose = verbose

    def get_action(self, env):
        if hasattr(env, "num_envs"):
            return env.full_like_actions(0.0)
        return np.zeros(env.number_of_ports)


# ---- agents that READ the env (single-env facade `ev2gym_amd.env.EV2Gym`, or the reference's own env: the same object graph) ----
def _ports(env):
    """(port number, charger, attached EV or None) in the reference's port order (charger by charger, port by port)."""
    n = 0
    for cs in env.charging_stations:
        connected = cs.evs_connected
        for j in range(cs.n_ports):
            yield n, cs, connected[j]
            n += 1


class RoundRobin:
    """heuristics.py:7-97: power-setpoint tracking by taking turns.  Every step ceil(setpoint / average charger power) EVs from
    the front of a queue of parked, not yet full EVs charge at full power (the last one takes the fractional remainder) and move
    to the back of the queue; newly parked EVs enter at the front."""
    algo_name = "Round Robin"

    def __init__(self, env, verbose=False, **kwargs):
        self.verbose, self.env = verbose, env
        total = 0
        for cs in env.charging_stations:
            total += cs.max_charge_current * cs.voltage * math.sqrt(cs.phases) / cs.n_ports
        self.average_power = total / len(env.charging_stations)
        self.number_of_ports_per_cs = env.number_of_ports_per_cs
        self.ev_buffer = []    # port numbers, next to be served first

    def get_env(self):
        return self.env

    def update_ev_buffer(self, env) -> None:
        for n, cs, ev in _ports(env):
            wants = ev is not None and ev.get_soc() < 1
            if wants and n not in self.ev_buffer:
                self.ev_buffer.insert(0, n)
            elif not wants and n in self.ev_buffer:
                self.ev_buffer.remove(n)

    def get_action(self, env):
        want = env.power_setpoints[env.current_step] * 1000 / self.average_power   # EVs' worth of power, in W / W
        self.update_ev_buffer(env)
        n = min(int(np.ceil(want)), len(self.ev_buffer))
        turn, self.ev_buffer = self.ev_buffer[:n], self.ev_buffer[n:]
        self.ev_buffer.extend(turn)
        act = np.zeros(env.number_of_ports)
        for i, port in enumerate(turn):
            act[port] = 1 / env.number_of_ports_per_cs
            if i == len(turn) - 1 and want < len(turn):
                act[port] = want - i
        return act


class ChargeAsLateAsPossible:
    """heuristics.py:100-149: an EV starts charging at full power at the last step from which it can still be full at departure."""
    algo_name = "Charge As Late As Possible"

    def __init__(self, verbose=False, **kwargs):
        self.verbose = verbose

    def get_action(self, env):
        act = np.zeros(env.number_of_ports)
        for n, cs, ev in _ports(env):
            if ev is None:
                continue
            power = min(cs.max_charge_current * cs.voltage * math.sqrt(cs.phases) / 1000, ev.max_ac_charge_power)
            steps_needed = math.ceil((1 - ev.get_soc()) / (power * env.timescale / 60 / ev.battery_capacity))
            if ev.get_soc() < 1 and ev.time_of_departure - steps_needed <= env.current_step:
                act[n] = 1
        return act


class ChargeAsFastAsPossibleToDesiredCapacity:
    """heuristics.py:230-267: full power until one more full step would overshoot the desired capacity, then the fraction that
    lands on it."""
    algo_name = "Charge As Fast As Possible To Desired Capacity"

    def __init__(self, verbose=False, **kwargs):
        self.verbose = verbose

    def get_action(self, env):
        act = np.zeros(env.number_of_ports)
        for n, cs, ev in _ports(env):
            if ev is None:
                continue
            cs_power = cs.get_max_power()
            step_energy = min(cs_power, ev.max_ac_charge_power) * env.timescale / 60
            if ev.current_capacity + step_energy < ev.desired_capacity:
                act[n] = 1
            else:
                act[n] = max(((ev.desired_capacity - ev.current_capacity) * 60 / env.timescale) / cs_power, 0)
        return act
