"""A/B switches of the host side.  Every `FSF_*` environment variable of DESIGN.md section 6 is read ONCE, here, when the package is
imported: the 900-launch frame pays attribute lookups instead of `os.environ` lookups, and a switch cannot change under a frame that
is running on another host thread.  Scratch scripts set the environment before importing the package
(`tools/profiling/ab_bench.sh "VAR=0" "VAR=1"` starts one process per setting); tests and bench.py that need both settings in one
process assign the attribute (`monkeypatch.setattr(switches, "TRAIN_PLANES", False)`).  Default = the fast path."""
import os


def _on(name):
    return os.environ.get(name, "1") != "0"


def _int(name, default):
    return int(os.environ.get(name, str(default)))


PLANES = _on("FSF_PLANES")                          # K9c / K9d dispatch (off: K9b / the fp32 kernel)
PLANES_STRIDED = _on("FSF_PLANES_STRIDED")          # ... also for strided / inverse convolutions
PLANES_MIN_ROWS = _int("FSF_PLANES_MIN_ROWS", 4096)
TRAIN_SPLIT = _on("FSF_TRAIN_SPLIT")                # training: K9b for forward / data gradient of the submanifold layers
TRAIN_PLANES = _on("FSF_TRAIN_PLANES")              # training: K9c / K9d wherever the direction's shape fits
TRAIN_BN = _on("FSF_TRAIN_BN")                      # training-mode BatchNorm (+ ReLU) on K23
SYNCBN_FUSED = _on("FSF_SYNCBN_FUSED")              # naiveSyncBN1d across ranks as one autograd node
UNET_LATERAL_STREAM = _on("FSF_UNET_LATERAL_STREAM")  # the fine lateral blocks on a side stream
UNET_LATERAL_LEVELS = _int("FSF_UNET_LATERAL_LEVELS", 3)
BOX_TAIL_FUSED = _on("FSF_BOX_TAIL_FUSED")            # inference: decode -> class ranks -> NMS -> selection as four C-ABI calls and one read-back
UNET_PLAN_STREAM = _on("FSF_UNET_PLAN_STREAM")        # inference: each level's rulebooks built one level ahead on a side stream
HEAD_SLICED = _on("FSF_HEAD_SLICED")                # the head's attribute branches as one sliced K22 launch per layer
SEG_HEAD_STACK = _on("FSF_SEG_HEAD_STACK")          # the segmentation head's two output Linears as one launch
SIR_SORTED = _on("FSF_SIR_SORTED")                  # inference: SIR stacks on rows sorted by group, segmented max fused into K22 (K22s)
SIR_GATHER = _on("FSF_SIR_GATHER")                  # first SIR layer reads the point features in place through an index
FUSED_VOTE = _on("FSF_FUSED_VOTE")                  # vote centres + cluster-voxel keys in one kernel
CLUSTER_ONE_UNIQUE = _on("FSF_CLUSTER_ONE_UNIQUE")  # a single unique in the cluster assignment
TRAIN_SIR_PRODUCT = _on("FSF_TRAIN_SIR_PRODUCT")    # training: SIRLayer's concatenations + product with the position MLP as one kernel each way (K28)
FUSION_ADD_FUSED = _on("FSF_FUSION_ADD_FUSED")      # inference: LiDAR + image point features summed in the epilogue of the update MLP's last Linear
GROUP_PAIRS = _on("FSF_GROUP_PAIRS")                # inference, one sample: the (group, point) pairs of the grouped sampling as one C-ABI call (K27)
REFINE_DIRECT = _on("FSF_REFINE_DIRECT")            # inference: the refine head's groups indexed by RoI directly (no unique, no scatter)
UNIQUE_BOUNDS = _on("FSF_UNIQUE_BOUNDS")            # uniques pack their sort key from bounds the key's producer attached (no range pass / host wait)
OVERLAP_ROWS = _on("FSF_OVERLAP_ROWS")              # inference: the camera-query row list (foreground + overlap duplicates) as two C-ABI calls (K26)
KEY_SURVIVAL = _on("FSF_KEY_SURVIVAL")              # ... and its density filter as one C-ABI call / one read-back (K25)
VFE_DECORATE = _on("FSF_VFE_DECORATE")              # the VFE input decoration in one kernel
LAZY_CAT = _on("FSF_LAZY_CAT")                    # inference: the U-Net decoder's channel concatenation is written only if read as one tensor
SPLIT_F16 = _on("FSF_SPLIT_F16")                  # inference: K9b's layers with whole 32-channel chunks on f16 x 3 planes (K9b-XP)
K22F = _on("FSF_K22F")                              # the <= 128-channel-slice Linears (K22 / K22s) on f16 x 3: x split in the kernel per row, W as f16 planes (K22f)
K22H = _on("FSF_K22H")                              # inference: the wide (>= 256 -> >= 256 channel) Linears on f16 x 3 planes (K22h)
K22H_MIN_ROWS = _int("FSF_K22H_MIN_ROWS", 1024)
UNET_MASK_ORDER = _on("FSF_UNET_MASK_ORDER")          # inference: the U-Net's fine levels in neighbour-mask row order
UNET_MASK_ORDER_LEVELS = _int("FSF_UNET_MASK_ORDER_LEVELS", 2)  # how many levels from the finest (1 = the input level only)
UNET_MASK_ORDER_MIN_ROWS = _int("FSF_UNET_MASK_ORDER_MIN_ROWS", 16384)
