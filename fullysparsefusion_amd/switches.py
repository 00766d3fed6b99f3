"""The host side's dispatch switches: the ELEVEN places where two built paths exist on purpose — a kernel family with a fallback of
a different shape, or a scheduling choice — each re-run in its off-setting by tests/test_switches_gpu.py against the default.  Read
ONCE from the environment when the package is imported (a frame pays attribute lookups, and a switch cannot change under a frame
that is running on another host thread); tests and bench.py assign the attribute (`monkeypatch.setattr(switches, "K22H", False)`).
Default = the fast path.

Round 6 removed 22 others: every "measured, not kept" experiment of rounds 2-5 had left a live branch behind an `FSF_*` variable.
Where the slower branch is ALSO the path for inputs the fast one does not cover (other dtypes, gradients, several samples), the branch
stays — selected by the input, not by a switch.  The row-count thresholds below are tuning constants, not switches."""
import os


def _on(name):
    return os.environ.get(name, "1") != "0"


PLANES = _on("FSF_PLANES")                            # sparse convolutions on pre-split f16 planes (K9d / K9c); off: K9b / the fp32 kernel
TRAIN_PLANES = _on("FSF_TRAIN_PLANES")                # training: K9c / K9d wherever the direction's shape fits; off: K9b both ways
SYNCBN_FUSED = _on("FSF_SYNCBN_FUSED")                # naiveSyncBN1d across ranks as one autograd node; off: the upstream formulation
UNET_LATERAL_STREAM = _on("FSF_UNET_LATERAL_STREAM")  # inference: the fine lateral blocks on a side stream
UNET_PLAN_STREAM = _on("FSF_UNET_PLAN_STREAM")        # inference: each level's rulebooks built one level ahead on a side stream
UNET_MASK_ORDER = _on("FSF_UNET_MASK_ORDER")          # inference: the U-Net's fine levels in neighbour-mask row order
BOX_TAIL_FUSED = _on("FSF_BOX_TAIL_FUSED")            # inference: decode -> class ranks -> NMS -> selection as four C-ABI calls, one read-back
SIR_SORTED = _on("FSF_SIR_SORTED")                    # inference: SIR stacks on rows sorted by group, segmented max fused into K22 (K22s)
REFINE_DIRECT = _on("FSF_REFINE_DIRECT")              # inference: the refine head's groups indexed by RoI directly (no unique, no scatter)
K22F = _on("FSF_K22F")                                # the <= 128-channel-slice Linears on f16 x 3 (x split in the kernel); off: bf16 x 6
K22H = _on("FSF_K22H")                                # the >= 256-wide Linears on f16 x 3 planes; off: K22 slices / the library

# ---- tuning constants (row counts from which a path pays; set once, measured in the round named) ----
PLANES_MIN_ROWS = 4096            # K9d from this many output rows (round 2)
K22H_MIN_ROWS = 1024              # K22h from this many rows (round 5)
UNET_LATERAL_LEVELS = 3           # how many fine levels' lateral blocks go to the side stream (round 3)
UNET_LATERAL_MIN_ROWS = 32768     # ... from this many voxels (round 6: the 1-sweep frame, ~20 k voxels, is 0.2-0.3 ms faster without it)
UNET_MASK_ORDER_LEVELS = 2        # how many levels from the finest run in neighbour-mask order (round 3)
UNET_MASK_ORDER_MIN_ROWS = 16384  # ... from this many voxels

ALL = ("PLANES", "TRAIN_PLANES", "SYNCBN_FUSED", "UNET_LATERAL_STREAM", "UNET_PLAN_STREAM", "UNET_MASK_ORDER", "BOX_TAIL_FUSED",
       "SIR_SORTED", "REFINE_DIRECT", "K22F", "K22H")
