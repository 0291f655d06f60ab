"""A SECOND full-depth LLaMA-7B gptq.int4 checkpoint against the reference itself (seed 1: other weights, scales and zero points than
tests/test_golden_7b_gpu.py's), prompt of 24, 32 greedy tokens: tests/golden/cfg2_7b_int4_s1.npz holds what the UNMODIFIED /root/reference
produced on the CPU (generate.py:63-91 with top_k = 1, then teacher-forced logits; oracle/gen_golden.py --big-s1, oracle == reference with
max |dlogit| = 0), cfg2_7b_int4_s1_bf16ref.npz the reference's own bf16 run on the same tokens (--big-bf16 --s1: 0.046-0.089 logit-std from
its f32 run).  north_star's "token-for-token" on more than one checkpoint (round-3 review: "one fixture pair is one fixture pair").
Same bars as the first checkpoint: teacher-forced logits within HALF the reference's own bf16 distance, argmax equal wherever the
reference's top-2 margin exceeds twice that (23 of the 32 steps), free-running greedy tokens equal up to the first step inside it; through the
persistent step (fp8 operands) and the launch-per-operator step.  The fixture was generated after round 4's GPU budget was spent: the file
sorts last so that its first GPU run cannot mask another test.
"""
import pytest
import torch

from test_golden_7b_gpu import _int4_checkpoint_against_its_fixtures

pytestmark = pytest.mark.gpu


@torch.no_grad()
def test_full_depth_7b_int4_second_checkpoint_against_the_reference_golden_run(dev, golden, record):
    _int4_checkpoint_against_its_fixtures(dev, golden, ("cfg2_7b_int4_s1",), record)
