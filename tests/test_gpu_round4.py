"""Round-4 GPU tests (through the C ABI): the kernels specialised for a model class.

`arb_step_kernel`'s FEAT bit 4 (csrc/arb_kernels.hip): for models with exactly four enabled plane / sphere SoftFingerContacts
(eight when the model has two column sets), no PD controller and no joint viscosity -- human36 on the floor, BASELINE
configs 3 and 5 and the reference's own eight-contact scenario -- the constraint type, the shape pair, nc and ndol are
compile-time constants and the code of the absent model features is not compiled in (+8 % on the headline workload).  Same expressions on the same values: the results must equal the general kernels' bit for bit, which
the flag ARB_STEP_GENERAL_KERNELS selects in the same library.
"""
import numpy as np
import pytest

from conftest import load_model
from arboris_python_amd import _capi

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu


def test_plan_reports_the_specialised_kernels(monkeypatch):
    from arboris_python_amd.batch import BatchedWorlds
    assert _capi.load().arb_build_variants() == 0
    # model: class bits in float32 (4: contacts, 8: no constraints, 4 | 16: contacts with body-space columns, 0: the general
    # kernels), in float64
    want = {"human36_c4": (52, 20),              # four plane / sphere SoftFingerContacts: body-space columns by default (round 6;
                                                 # float32: compiled for four contacts, FEAT bit 32)
            "human36_c8": (20, 20),              # eight on two feet: body-space constraint columns, ONE column set (round 5)
            "human36_g": (8, 0),                 # no constraints (BASELINE config 2): float32 only
            "human36_visc": (0, 0),              # joint viscosity: outside the classes
            "human36_c4_pdw": (0, 0),            # a PD controller in the model: a dense impedance, outside the classes
            "simplearm": (0, 0)}
    for name, (spec32, spec64) in want.items():
        m, _, _ = load_model(name)
        bw = BatchedWorlds(m)
        for B, T in ((512, 1), (8192, 40)):
            assert bw.plan(B, T)["feat"] == spec32, (name, B, T)
            assert bw.plan(B, T, ext_gforce=True)["feat"] == (spec32 | 1), (name, B, T)
            # every optional input: the general kernel (with body-space columns when the model has them)
            assert bw.plan(B, T, other_inputs=True)["feat"] == (19 if spec32 & 16 else 3), (name, B, T)
            assert bw.plan(B, T, dtype=torch.float64)["feat"] == spec64, (name, B, T)
        if spec32:
            assert bw.plan(8192, 40, general_kernels=True)["feat"] == 0
        if name == "human36_c4":                 # the classical columns on request: the kernels specialised for four contacts
            assert bw.plan(8192, 40, classic_columns=True)["feat"] == 4 and bw.plan(8192, 40, dtype=torch.float64, classic_columns=True)["feat"] == 4
            assert bw.plan(8192, 40, ext_gforce=True, classic_columns=True)["feat"] == 5
        bw.close()


@pytest.mark.parametrize("torques", [False, True])
@pytest.mark.parametrize("model,dtype", [("human36_c4", "float32"), ("human36_c4", "float64"), ("human36_g", "float32")])
def test_specialised_kernels_equal_the_general_ones_bitwise(monkeypatch, model, dtype, torques):
    """Whole falling episodes (free fall, impact, sliding, the rare routes of the local solve late in the episode), two- and
    three-wave builds, the work queue, one launch per step; plain inputs (FEAT 4 against 0) and user torques (5 against 1);
    float32 and float64 with four contacts; the class without constraints (FEAT 8 / 9: human36 in free motion, BASELINE
    config 2).  (Eight contacts: the body-space-column kernels of round 5 equal the general two-column-set kernels to
    rounding, not bit for bit -- tests/test_gpu_round5.py.)"""
    from arboris_python_amd import synth
    from arboris_python_amd.batch import BatchedWorlds
    m, _, _ = load_model(model)
    dt_ = getattr(torch, dtype)
    bw = BatchedWorlds(m)
    T = 40
    cases = [(4096, "episode", {}), (700, "episode", {}), (1500, "per_step", {})]
    if model == "human36_c4" and dtype == "float32":
        cases.insert(1, (4096, "episode", dict(waves=2)))
    for B, mode, kw in cases:
        q, dq = synth.standing_states(m, B, seed=4000 + B, drop=0.03, vel=0.1)
        ext = None
        if torques:
            rng = np.random.default_rng(B)
            ext = torch.as_tensor(rng.normal(0., 0.5, (B, m.ndof)), dtype=dt_, device="cuda")
        cls = 8 if m.nc == 0 else 4
        # (classic_columns: the specialised kernels of the four-contact class run the CLASSICAL columns; the model's default since
        # round 6 are body-space columns, equal to these to rounding only -- tests/test_gpu_round5.py)
        assert bw.plan(B, 1 if mode == "per_step" else T, dtype=dt_, ext_gforce=torques, classic_columns=True, **kw)["feat"] == (cls | (1 if torques else 0))
        res = {}
        for key in ("spec", "general"):
            gk = dict(general_kernels=(key == "general"), classic_columns=True)
            tq, tdq = bw.to_device(q, dq, dt_)
            cf = bw.new_cforce(B, dt_) if m.nc else None
            if mode == "per_step":
                for _ in range(T):
                    bw.step(tq, tdq, 5e-3, 1, cforce=cf, ext_gforce=ext, **gk)
            else:
                bw.step(tq, tdq, 5e-3, T, cforce=cf, ext_gforce=ext, **kw, **gk)
            torch.cuda.synchronize()
            bw.status()
            res[key] = (tq, tdq) if cf is None else (tq, tdq, cf)
        if not torques:
            assert bool(torch.isfinite(res["spec"][0]).all())
        if m.nc:
            assert float(torch.nan_to_num(res["spec"][2]).abs().max()) > 0.        # (the contacts did act)
        bits = torch.int32 if dtype == "float32" else torch.int64                   # (bit patterns: a diverged world's NaN included)
        assert all(torch.equal(a.view(bits), b.view(bits)) for a, b in zip(res["spec"], res["general"])), (B, mode, kw)
    bw.close()
