"""One switch for the arithmetic of every MLP kernel.

  ROBIR_PRECISION=exact  (default)  not narrower than the reference's fp32: the fused light-visibility kernel carries every
                                    fp32 operand exactly as three f16 pieces (six MFMA products per multiply-add, three fp32
                                    accumulators: csrc/vis_diffuse_x6.hip; since round 6 the two outer products of the 2^-22 class from bf8
                                    copies of their operands, error against float64 unchanged: csrc/vis_diffuse_x6t.hip XT_FP8), and so do the SDF network (values and reverse-mode
                                    gradient), the colour network, the stand-alone visibility MLP, the 512-wide nets and the CESR
                                    nets (MLP mode "f16x6": csrc/sdf_x6.hip, sdf_back_x6.hip, color_x6.hip, vis_x6.hip, wide_x6.hip,
                                    cesr_x6.hip); the small auto-encoder decoders run on the f32-input MFMA.  ROBIR_MLP_PRECISION=fp32 puts every stand-alone MLP on the f32-input MFMA;
  ROBIR_PRECISION=split             22-bit operands: (hi, lo) f16 pairs, three products per multiply-add, fp32 accumulate --
                                    2x the throughput, parity-tested against the oracle at the same 1e-4 (tests/test_precision_gpu.py),
                                    guarded by the activation-range sentinel (ops.range_check).
  ROBIR_PRECISION=f16               the labelled THROUGHPUT policy `BASELINE.json configs[4]` names ("fp16 MLP weights on MFMA"):
                                    the light-visibility MLP in PLAIN f16 -- one f16 MFMA product per multiply-add, f16 weights
                                    (round-to-nearest) and f16 activations (truncated between the layers), fp32 accumulation
                                    (csrc/vis_diffuse_f16p.hip, vis_diffuse_f16t.hip) -- the two CESR nets likewise since round 6
                                    (csrc/cesr_f16.hip: 52 % of config 5) -- and, since round 5, every other net in SPLIT
                                    precision (f16 hi/lo weight and activation pairs, three products: the parity-tested family of
                                    `split`, legacy library) instead of the exact-operand kernels: no one-product kernels exist for
                                    those nets, and a throughput policy that left them at six products gave the CESR stage 18 %.
                                    NARROWER than the reference's fp32: never a default, never a parity claim; its measured
                                    error against a float64 evaluation is in DESIGN.md and printed by tests/test_precision_gpu.py.
  ROBIR_PRECISION=f16-vis           round 4's meaning of `f16`, kept selectable: plain f16 for the light-visibility MLP only,
                                    every other net on the exact-operand kernels -- needs the default library only.
ROBIR_VIS_PRECISION / ROBIR_MLP_PRECISION override the two halves of the policy separately (A/B runs, tests).
"""
import os

POLICIES = {"exact": ("f16x6", "f16x6"), "split": ("f16x3-auto", "f16x3"), "f16": ("f16x1", "f16x3"), "f16-vis": ("f16x1", "f16x6")}
VIS_MODES = ("fp32", "f16x6", "f16x3-auto", "f16x3-v3", "f16x3-v2", "f16x3", "f16x1", "f16x6-1t", "f16x6-pt", "f16x6-stream")


def policy():
    p = os.environ.get("ROBIR_PRECISION", "exact")
    if p not in POLICIES:
        raise ValueError("ROBIR_PRECISION must be exact, split, f16 or f16-vis")
    return p


def mlp_precision():
    """Arithmetic of the stand-alone MLP kernels (SDF, colour, visibility, 512-wide nets): 'f16x6' (exact three-piece operands where
    such a kernel exists -- SDF, colour, visibility MLP -- and the f32-input MFMA elsewhere), 'fp32' (f32-input MFMA everywhere) or
    'f16x3' (split precision)."""
    p = os.environ.get("ROBIR_MLP_PRECISION") or POLICIES[policy()][1]
    if p not in ("f16x3", "fp32", "f16x6"):
        raise ValueError("ROBIR_MLP_PRECISION must be f16x6, fp32 or f16x3")
    return p


def cesr_precision():
    """Arithmetic of the two CESR nets (shadow_net, normal_net: training/train_cesr.py:106-110): 'f16x1' -- plain f16, ONE MFMA product per
    multiply-add (csrc/cesr_f16.hip, round 6; NARROWER than fp32) -- under ROBIR_PRECISION=f16, the policy `BASELINE.json configs[4]` names
    ("CESR stage full pipeline, fp16 MLP weights on MFMA"); else whatever mlp_precision() says.  ROBIR_CESR_PRECISION overrides (A/B runs)."""
    p = os.environ.get("ROBIR_CESR_PRECISION")
    if p is None:
        p = "f16x1" if (policy() == "f16" and not os.environ.get("ROBIR_MLP_PRECISION")) else mlp_precision()
    if p not in ("f16x1", "f16x3", "fp32", "f16x6"):
        raise ValueError("ROBIR_CESR_PRECISION must be f16x1, f16x6, fp32 or f16x3")
    return p


def vis_precision():
    """Arithmetic of the fused light-visibility kernel at import time of robir_amd.sg_render (its VIS_PRECISION attribute is
    what is read per call)."""
    p = os.environ.get("ROBIR_VIS_PRECISION") or POLICIES[policy()][0]
    if p not in VIS_MODES:
        raise ValueError("ROBIR_VIS_PRECISION must be one of " + ", ".join(VIS_MODES))
    return p
