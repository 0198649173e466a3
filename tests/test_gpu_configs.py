"""-m gpu: the BASELINE.json configurations at their own workload against the
oracle, EVERY instance of the batch replayed (the oracle runs one Solver per
instance on host threads, as the reference runs one per worker):

* configs[2]  SPP aa04-shaped (equalities-01, `synth.config_c3`), 64 restarts,
* configs[4]  telebus-shaped (inequalities-Z, `synth.config_c5`), 64 and 256
  restarts in one batch,
* configs[1]  scpnrh1-shaped (`synth.config_c2`): the exact `bench.py` step
  sequence (reset every 8th restart, x <- Bastert point, compute_order::init,
  one sweep with the kappa of best_solution_recorder::reinit's cycle) for 16
  restart-sweeps, fp64 and fp32, launched back to back as the bench does, and
  a soak of 200 steps; also with a batch that is not a multiple of 8, so that
  the four workgroups of an instance sit on different XCDs.

Bit-exact on x, pi, P, remaining and the engine state (tolerance 0).
"""
import os
from concurrent.futures import ThreadPoolExecutor

import numpy as np
import pytest

from baryonyx_amd import synth
from baryonyx_amd.driver import Params, compute_delta, normalize_costs
from helpers import bits_equal, first_diff

pytestmark = pytest.mark.gpu


def host_threads():
    n = len(os.sched_getaffinity(0))
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(float(quota) / float(period) + 0.5)))
    except OSError:
        pass
    return max(1, min(n, 32))


def on_threads(fn, items):
    """fn(item) for every item on the host's cores (ctypes calls drop the GIL)."""
    with ThreadPoolExecutor(max_workers=host_threads()) as ex:
        return list(ex.map(fn, items))


def kappa_cycle(params, k):  # bench.py's (best_solution_recorder::reinit, kappa-improve)
    steps = int(round((params.init_kappa_improve_stop - params.init_kappa_improve_start)
                      / params.init_kappa_improve_increase))
    app = params.init_kappa_improve_start + params.init_kappa_improve_increase * (1 + k % max(steps, 1))
    return params.kappa_min + (params.kappa_max - params.kappa_min) * app


def check_all(dev, orcs, st, tag, with_P=True):
    for b, s in enumerate(orcs):
        t = f"{tag} inst {b}"
        assert np.array_equal(dev.get_x(b), s.get_x()), t + " x"
        pd, po = dev.get_pi(b), s.get_pi()
        assert bits_equal(pd, po), f"{t} pi first diff at {first_diff(pd, po)}"
        assert st["rng"][b] == s.rng, t + " rng"
        assert st["row_updates"][b] == s.row_updates, t + " row_updates"
        if with_P:
            Pd, Po = dev.get_P(b), s.get_P()
            assert bits_equal(Pd, Po), f"{t} P first diff at {first_diff(Pd, Po)}"


def run_config(oracle, bxh, inst, batch, float_type, sweeps, push_at, order=0):
    """solver_functor-like sequence of sweeps, every instance with its own
    engine and kappa, all of them replayed."""
    theta = 0.5
    orcs = on_threads(lambda b: oracle.Solver(inst, float_type=float_type, sort_mode=1, cost_norm=4,
                                              seed=3000 + b), range(batch))
    cost = orcs[0].cost()
    delta = orcs[0].compute_delta(theta)
    dev = bxh.DeviceBatch(inst, cost, batch, float_type=float_type)
    x0 = np.zeros(dev.nwords, np.uint64)
    for b, s in enumerate(orcs):
        s.set_x(x0)
        dev.set_x(b, x0)
        dev.set_rng(b, s.rng)
        s.order_init(order)
    rem = dev.order_init(order)
    assert [int(r) for r in rem] == [len(s.violated()) for s in orcs]
    kappa = np.linspace(0.0, 0.2, batch)
    for it in range(sweeps):
        push = it in push_at
        amp = 5.0 if push else 0.0
        st = dev.sweep(kappa, delta, theta, is_push=push, obj_amp=amp, order=order)
        rs = on_threads(lambda b: orcs[b].sweep(float(kappa[b]), delta, theta, is_push=push,
                                                obj_amp=amp), range(batch))
        assert [int(r) for r in st["remaining"]] == rs, f"sweep {it} remaining"
        check_all(dev, orcs, st, f"sweep {it}")
        kappa = kappa + 0.03
    form = dev.kernel_form
    dev.close()
    return form


def test_config_c3_spp_matches_oracle(oracle, bxh):
    """BASELINE configs[2]: aa04-shaped set partitioning, equalities-01
    (itm-solver-equalities-01.cpp:184-270), 64 restarts."""
    inst = synth.config_c3("int")
    assert inst.kind == 0
    run_config(oracle, bxh, inst, 64, 1, sweeps=8, push_at=(4, 6))


def test_config_c3_spp_f32_random_order(oracle, bxh):
    inst = synth.config_c3("int")
    run_config(oracle, bxh, inst, 64, 0, sweeps=5, push_at=(3,), order=2)


@pytest.mark.parametrize("batch", [64, 256])
def test_config_c5_z_matches_oracle(oracle, bxh, batch):
    """BASELINE configs[4]: telebus-shaped general integer coefficients,
    inequalities-Z (itm-solver-inequalities-Z.cpp:441-564), 64 restarts and the
    256 restarts per GPU the config names."""
    inst = synth.config_c5("int")
    assert inst.kind == 4
    form = run_config(oracle, bxh, inst, batch, 1, sweeps=5 if batch == 256 else 8, push_at=(3,))
    assert form in (3, 6), form  # 6: the wave-per-row form of the Z kernel (linear objective)


def bench_steps(oracle, bxh, batch, float_type, steps, p_every, expect_form=44, cost_kind="int"):
    """bench.py's step, `steps` times, launched back to back; every instance is
    replayed by the oracle.  x, pi, engine, remaining are compared after every
    step; P every `p_every` steps and at both sides of a reset."""
    inst = synth.config_c2(cost_kind)
    params = Params()
    cost = normalize_costs(inst.dense_costs(), params.cost_norm, float_type)
    delta = compute_delta(cost, params.theta, float_type)
    dev = bxh.DeviceBatch(inst, cost, batch, float_type=float_type)
    if expect_form is not None:
        assert dev.kernel_form == expect_form, dev.kernel_form
    orcs = on_threads(lambda b: oracle.Solver(inst, float_type=float_type, sort_mode=1, cost_norm=4,
                                              seed=1), range(batch))
    assert bits_equal(orcs[0].cost(), cost)
    assert orcs[0].compute_delta(params.theta) == delta
    for b, s in enumerate(orcs):
        seed = 1 + (7919 + b * 104729) % 2147483645  # bench.py's engines of rank 0
        s.rng = seed
        dev.set_rng(b, seed)
    x0 = np.zeros(dev.nwords, np.uint64)  # the Bastert point of a min-cost covering instance
    x0all = np.tile(x0, (batch, 1))

    def oracle_step(b, k):
        s = orcs[b]
        if k % 8 == 0:
            s.reset()
        s.set_x(x0)
        s.order_init(params.order)
        return s.sweep(kappa_cycle(params, k), delta, params.theta)

    for k in range(steps):
        if k % 8 == 0:
            dev.reset_all()
        dev.set_x_all(x0all)
        dev.order_init(params.order, sync=False)
        dev.sweep_async(np.full(batch, kappa_cycle(params, k)), delta, params.theta, order=params.order)
        rs = on_threads(lambda b: oracle_step(b, k), range(batch))  # while the GPU runs
        st = dev.status()
        assert [int(r) for r in st["remaining"]] == rs, f"step {k} remaining"
        with_P = (k % p_every == 0) or (steps <= 16 and k % 8 in (0, 7)) or k == steps - 1
        check_all(dev, orcs, st, f"step {k}", with_P=with_P)
    dev.close()


@pytest.mark.parametrize("float_type", [1, 0], ids=["f64", "f32"])
def test_bench_steps_all_instances_match_oracle(oracle, bxh, float_type):
    """configs[1] as bench.py runs it: 64 instances, 16 restart-sweeps (two
    resets), the pipelined sweep (row chain + 3 producers per instance)."""
    bench_steps(oracle, bxh, 64, float_type, steps=16, p_every=4)


def test_bench_steps_partners_on_different_xcds(oracle, bxh):
    """61 instances: 4 * 61 workgroups, so blocks b .. b + 3 of an instance are
    dealt to different XCDs (the P / pi hand-over then crosses L2s)."""
    bench_steps(oracle, bxh, 61, 1, steps=9, p_every=4)


def test_bench_steps_real_costs(oracle, bxh):
    """tie-free costs: no row takes the sort path after the first sweep."""
    bench_steps(oracle, bxh, 64, 1, steps=9, p_every=4, cost_kind="real")


def test_bench_steps_soak(oracle, bxh):
    """200 bench steps (25 solves), every instance checked after every step."""
    bench_steps(oracle, bxh, 64, 1, steps=200, p_every=10)
