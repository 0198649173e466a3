"""-m gpu: baryonyx::optimize of the C++ host library (the thread pool of
optimize_functor workers as one batch of solver instances on the GPU,
baryonyx_amd/host/itm_gpu.cpp) against the oracle's optimize() under the same
deterministic schedule (oracle/itm_oracle_opt.hpp), after a fixed number of
rounds: identical incumbent, identical population (remaining constraints,
value, hash and x of every entry, best first), identical engine state and
restart count of every worker.  Rows a17 / a18 of SURVEY.md 8(a); and, through
solve(), the x initialisers (a19: pessimistic / optimistic pre-solve) and the
cost norms (a15: none / l1 / l2 / random) that no other GPU test sets."""
import numpy as np
import pytest

from baryonyx_amd import host_ffi as H
from baryonyx_amd import synth
from helpers import pack_bits

pytestmark = pytest.mark.gpu

COMMON = ("theta delta kappa_min kappa_step kappa_max alpha pushing_k_factor "
          "pushing_objective_amplifier init_policy_random w limit seed pushes_limit "
          "pushing_iteration_limit init_population_size order init_policy cost_norm "
          "init_crossover_bastert_insertion init_crossover_solution_selection_mean "
          "init_crossover_solution_selection_stddev init_mutation_variable_mean "
          "init_mutation_variable_stddev init_mutation_value_mean init_mutation_value_stddev "
          "init_kappa_improve_start init_kappa_improve_increase init_kappa_improve_stop").split()

OPT_CASES = [
    # (name, maker, float_type, thread, rounds, params)
    ("scp", lambda: synth.make_scp(60, 400, 0.06, "int", 12), 1, 8, 300,
     dict(limit=100, w=5.0, init_population_size=20)),
    ("scp-f32-random-order", lambda: synth.make_scp(50, 300, 0.08, "int", 15), 0, 6, 200,
     dict(limit=60, w=3.0, init_population_size=12, order=2, pushes_limit=4, pushing_iteration_limit=6)),
    ("spp", lambda: synth.make_spp(40, 300, 0.06, "int", 17), 1, 8, 300,
     dict(limit=80, w=5.0, init_population_size=16, kappa_max=0.9)),
    ("pm1-max", lambda: synth.make_pm1(50, 250, 0.06, "real", 20), 1, 5, 250,
     dict(limit=80, w=5.0, init_population_size=20, pushes_limit=3, pushing_iteration_limit=10)),
    ("pm1-eq-infeasibility-order", lambda: synth.make_pm1(40, 250, 0.06, "int", 21, equalities=True), 1, 4, 200,
     dict(limit=50, w=5.0, init_population_size=10, order=3)),
    ("knapsack-z", lambda: synth.make_z(20, 60, 0.2, "int", 11), 1, 6, 200,
     dict(limit=50, w=5.0, init_population_size=12, pushes_limit=2, pushing_iteration_limit=5)),
    ("scp-l2-norm", lambda: synth.make_scp(60, 400, 0.06, "real", 13), 1, 4, 150,
     dict(limit=60, w=5.0, init_population_size=10, cost_norm=3)),
    # every restart is a crossover (init_kappa_improve_stop = 0) followed by a mutation
    # that redraws a few per cent of the variables: the restarts of the device-resident
    # population (bxh_reinit: masks and the mutation loop drawn on the device)
    ("crossover-every-restart", lambda: synth.make_scp(60, 400, 0.06, "int", 41), 1, 8, 300,
     dict(limit=25, w=5.0, init_population_size=12, init_kappa_improve_stop=0.0,
          init_mutation_variable_mean=0.02, init_mutation_variable_stddev=0.01)),
    # ... mostly through the Bastert / random-vector branches of crossover()
    # (itm-optimizer-common.hpp:377-407), which also overwrite population slots
    ("crossover-insertions", lambda: synth.make_spp(40, 300, 0.06, "int", 42), 1, 6, 300,
     dict(limit=20, w=5.0, init_population_size=10, init_kappa_improve_stop=0.0,
          init_crossover_bastert_insertion=0.7, kappa_max=0.9)),
    ("crossover-no-mutation-f32", lambda: synth.make_pm1(50, 250, 0.06, "real", 43), 0, 5, 250,
     dict(limit=20, w=5.0, init_population_size=10, init_kappa_improve_stop=0.04,
          init_mutation_value_mean=0.0, init_mutation_value_stddev=0.0)),
    # cost_norm = random: every worker normalises with its OWN engine and so solves with its
    # own cost vector and its own automatic delta (itm-optimizer-common.hpp:633-638,
    # make_random_norm itm-common.hpp:1025-1075) -- one handle per worker on the device side
    ("random-norm-per-worker", lambda: synth.make_scp(60, 400, 0.06, "int", 51), 1, 6, 250,
     dict(limit=60, w=5.0, init_population_size=12, cost_norm=1)),
    ("random-norm-per-worker-pm1-f32", lambda: synth.make_pm1(50, 250, 0.06, "int", 52), 0, 4, 200,
     dict(limit=50, w=5.0, init_population_size=10, cost_norm=1, delta=0.01, pushes_limit=3,
          pushing_iteration_limit=6)),
]


def both_params(oracle, **kw):
    hp = H.default_params(**kw)
    okw = {k: getattr(hp, k) for k in COMMON}
    return hp, okw


@pytest.mark.parametrize("reinit", ["device", "host"])
@pytest.mark.parametrize("case", OPT_CASES, ids=[c[0] for c in OPT_CASES])
def test_optimize_matches_oracle(oracle, tmp_path, case, reinit, monkeypatch):
    """`device`: the population lives on the device and a restart's crossover /
    mutation run there (bxh_reinit; at the end the device's copy of every slot is
    compared with the host's, BXC_CHECK_POP); `host`: x is built on the host and
    uploaded (BXC_HOST_REINIT).  Both must be the oracle's run."""
    monkeypatch.setenv("BXC_CHECK_POP", "1")
    if reinit == "host":
        monkeypatch.setenv("BXC_HOST_REINIT", "1")
    else:
        monkeypatch.delenv("BXC_HOST_REINIT", raising=False)
    name, maker, float_type, thread, rounds, params = case
    path = str(tmp_path / "p.lp")
    synth.write_lp(maker(), path)
    inst = H.merged(path)  # what baryonyx::optimize hands to its solvers (after the preprocessor)
    hp, okw = both_params(oracle, seed=4242, thread=thread, float_type=float_type, sweep_budget=rounds,
                          time_limit=600.0, **params)
    res, tr = H.optimize_trace(path, hp, inst.n)
    ores, otr = oracle.optimize(inst, float_type, oracle.default_params(sort_mode=1, **okw),
                                thread=thread, rounds=rounds)
    assert tr["rounds"] == rounds == ores.sweeps
    assert np.array_equal(tr["worker_calls"], otr["worker_calls"])
    assert np.array_equal(tr["worker_rng"], otr["worker_rng"])
    assert np.array_equal(tr["pop_remaining"], otr["pop_remaining"])
    assert np.array_equal(tr["pop_value"], otr["pop_value"])
    assert np.array_equal(tr["pop_hash"], otr["pop_hash"])
    assert np.array_equal(tr["pop_x"], otr["pop_x"])
    assert res.remaining_constraints == ores.remaining_constraints
    assert res.loop == ores.loop
    if ores.status == 0:
        assert res.status == 0 and res.valid == 1
        assert res.value == ores.value == res.recomputed_value
    else:
        assert res.status == 3  # time_limit_reached: no feasible point in the budget


SOLVE_VARIANTS = [
    ("pessimistic-pre-solve", dict(init_policy=1)),
    ("optimistic-pre-solve", dict(init_policy=2, init_policy_random=0.1)),
    ("norm-none", dict(cost_norm=0)),
    ("norm-random", dict(cost_norm=1)),
    ("norm-l1", dict(cost_norm=2)),
    ("norm-l2", dict(cost_norm=3, init_policy=1)),
]


@pytest.mark.parametrize("maker", [lambda: synth.make_scp(60, 400, 0.06, "int", 31),
                                   lambda: synth.make_pm1(50, 250, 0.06, "real", 32)],
                         ids=["scp", "pm1"])
@pytest.mark.parametrize("name,params", SOLVE_VARIANTS, ids=[v[0] for v in SOLVE_VARIANTS])
def test_solve_init_policies_and_cost_norms_match_oracle(oracle, tmp_path, maker, name, params):
    """init_with_pre_solve (itm-common.hpp:284-374) and the cost norms of
    default_cost_type (:1000-1148; the random norm draws from the solve's own
    engine) through baryonyx::solve == the oracle's solve()."""
    path = str(tmp_path / "p.lp")
    synth.write_lp(maker(), path)
    inst = H.merged(path)
    hp, okw = both_params(oracle, seed=97531, limit=400, w=10.0, pushes_limit=3, pushing_iteration_limit=8,
                          **params)
    res, x = H.run(path, 0, hp)
    ores, oxw = oracle.solve(inst, 1, oracle.default_params(sort_mode=1, **okw))
    assert ores.status == res.status
    # (solve()'s result keeps remaining_constraints at its default, INT_MAX:
    # itm-solver-common.hpp:216-224)
    if res.status == 0:
        assert res.valid == 1 and ores.value == res.value == res.recomputed_value
        assert np.array_equal(pack_bits(x[: inst.n].astype(np.uint8)), oxw)


def test_optimize_over_several_devices_is_the_same_run(oracle, tmp_path, monkeypatch):
    """baryonyx::optimize deals its workers to the devices of
    context_set_parameters("devices", ...) round robin (one host recorder shared
    by all of them: no collective on the path): the schedule does
    not depend on where a worker lives, so three "devices" (here all device 0: one
    batch of instances each) give the run of one device, which is the oracle's."""
    path = str(tmp_path / "p.lp")
    synth.write_lp(synth.make_scp(60, 400, 0.06, "int", 12), path)
    inst = H.merged(path)
    hp, okw = both_params(oracle, seed=99, thread=8, float_type=1, sweep_budget=250, time_limit=600.0,
                          limit=100, w=5.0, init_population_size=20)
    monkeypatch.delenv("BXC_DEVICES", raising=False)
    res1, tr1 = H.optimize_trace(path, hp, inst.n)
    monkeypatch.setenv("BXC_DEVICES", "0,0,0")
    res3, tr3 = H.optimize_trace(path, hp, inst.n)
    for k in tr1:
        assert np.array_equal(tr1[k], tr3[k]), k
    assert (res1.value, res1.loop, res1.remaining_constraints) == (res3.value, res3.loop, res3.remaining_constraints)
    ores, otr = oracle.optimize(inst, 1, oracle.default_params(sort_mode=1, **okw), thread=8, rounds=250)
    assert np.array_equal(tr3["pop_hash"], otr["pop_hash"]) and np.array_equal(tr3["worker_rng"], otr["worker_rng"])


def test_random_cost_norm_gives_every_worker_its_own_costs(oracle, tmp_path):
    """With integer costs (many equal ones) make_random_norm spreads the equal
    costs with draws from the worker's engine: two workers do not hold the same
    vector, and the run differs from one that would share worker 0's."""
    path = str(tmp_path / "p.lp")
    synth.write_lp(synth.make_scp(40, 200, 0.08, "int", 5), path)
    inst = H.merged(path)
    hp, okw = both_params(oracle, seed=3, thread=4, cost_norm=1, sweep_budget=120, time_limit=600.0, limit=40,
                          w=5.0, init_population_size=10)
    res, tr = H.optimize_trace(path, hp, inst.n)
    ores, otr = oracle.optimize(inst, 1, oracle.default_params(sort_mode=1, **okw), thread=4, rounds=120)
    for k in ("worker_calls", "worker_rng", "pop_remaining", "pop_value", "pop_hash", "pop_x"):
        assert np.array_equal(tr[k], otr[k]), k
    res1, _ = H.run(path, 1, H.default_params(seed=3, thread=1, cost_norm=1, sweep_budget=50, time_limit=5.0))
    assert res1.status in (0, 3)
