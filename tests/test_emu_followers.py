"""Workgroups of one launch handing data to each other (round 6): the panel solve FOLLOWING the diagonal block through progress
words -- inside the single-theta step kernel (potrf_follow) and as the merged diagonal-block + panel launch of batched fits
(potrf_batch_follow).  Through the interpreter: it runs the workgroups of a launch in index order, the diagonal workgroup(s)
first, so what is checked here is the column-oriented substitution, the publication's indexing and the tile bookkeeping of
the reshuffled step kernel -- bit for bit against the launch-per-phase form.  The hand-off itself (write-through stores,
polls, L1-bypassing loads across XCDs) is what the MI355X runs of the same checks are for
(tests/test_gpu_parity.py::test_panel_followers_hand_off, ::test_batched_followers_hand_off)."""
import os
import sys

import pytest

import parity_checks as P
from robo_amd import _lib

HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.fixture(scope="module")
def emu_ctx():
    sys.path.insert(0, os.path.join(HERE, "hipemu"))
    import build_emu
    _lib.use_library(build_emu.build())
    ctx = _lib.Context(0)
    yield ctx
    ctx.close()
    _lib.use_library(None)


def test_panel_followers(emu_ctx):
    """the follower form of the single-theta factorisation: same bits as the launch-per-phase form, whichever step the
    hand-off starts at, however many workgroups share the other tiles"""
    P.check_panel_followers(emu_ctx, sizes=((512, 3), (300, 2)), froms=(-1, 1))


def test_batched_followers(emu_ctx):
    """the merged diagonal-block + panel launch of the batched factorisation: same likelihoods, kept factors, posteriors"""
    P.check_batched_followers(emu_ctx, sizes=((512, 3, 4),), groups=(0, 1))
