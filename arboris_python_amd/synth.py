"""Seeded synthetic states for the benchmark/parity configurations.

All generators return float64 ``(q, dq)`` with shapes (B, nq) and (B, ndof) in
the flat state layout of ``flatten.flatten_world`` (FreeJoint pose = 16 scalars,
row-major 4x4).
"""
import numpy as np

from . import homogeneousmatrix as Hg
from .flatten import JT_FREE, JT_TXTYTZ


def random_states(model, B, seed=0, angle=1.0, vel=3.0, root_box=((-1., 1.), (0.5, 1.5), (-1., 1.)),
                  root_rot=True):
    """Config-2/4 style states: hinge angles U(-angle, angle), velocities
    U(-vel, vel), every FreeJoint pose = transl(U) . rotzyx(U(-pi,pi), U(-1,1), U(-1,1))."""
    rng = np.random.default_rng(seed)
    q = np.zeros((B, model.nq))
    dq = rng.uniform(-vel, vel, size=(B, model.ndof))
    for b in range(model.nb):
        qs = slice(int(model.q_off[b]), int(model.q_off[b] + model.jnq[b]))
        if model.jtype[b] == JT_FREE:
            for w in range(B):
                t = [rng.uniform(lo, hi) for (lo, hi) in root_box]
                H = Hg.transl(*t)
                if root_rot:
                    H = H @ Hg.rotzyx(rng.uniform(-np.pi, np.pi), rng.uniform(-1, 1),
                                      rng.uniform(-1, 1))
                q[w, qs] = H.ravel()
        else:
            q[:, qs] = rng.uniform(-angle, angle, size=(B, int(model.jnq[b])))
    return q, dq


def standing_states(model, B, seed=0, drop=0.03, vel=0.1):
    """Config-3/5 states: q = 0, root lifted by U(0, drop) along +y, small
    random velocities U(-vel, vel) (the reference scenario of
    tests/test_human36_falling.py:10-17 uses drop = 0.03 exactly, zero velocity)."""
    rng = np.random.default_rng(seed)
    q = np.zeros((B, model.nq))
    dq = rng.uniform(-vel, vel, size=(B, model.ndof))
    for b in range(model.nb):
        if model.jtype[b] == JT_FREE:
            qs = slice(int(model.q_off[b]), int(model.q_off[b] + 16))
            y = rng.uniform(0., drop, size=B)
            H = np.tile(np.eye(4), (B, 1, 1))
            H[:, 1, 3] = y
            q[:, qs] = H.reshape(B, 16)
    return q, dq


def world_states(model, worlds, kind="standing", seed=0, **kw):
    """States of the worlds with GLOBAL indices ``worlds`` (an iterable of ints) of a seeded batch in which world w
    is drawn from its own stream ``default_rng([seed, w])`` ("seed = rollout index"): whatever range of
    the batch a rank takes, and however large the batch is, world w is the same world -- so a 1-GPU run and rank 0 of
    an 8-GPU run step identical worlds.  ``kind``: "standing" (standing_states) or "random" (random_states)."""
    gen = standing_states if kind == "standing" else random_states
    worlds = list(worlds)
    q = np.zeros((len(worlds), model.nq)); dq = np.zeros((len(worlds), model.ndof))
    for i, w in enumerate(worlds):
        q[i], dq[i] = (a[0] for a in gen(model, 1, seed=[int(seed), int(w)], **kw))
    return q, dq
