// The attention kernels of csrc/attention.hip instantiated for key width != value width: (128, 64) and (256, 128),
// the shapes of multi-head latent attention with decoupled rotary keys in its per-head (not absorbed) form
// (reference: single-gpu/model.py:237-345 with the launch configuration of single-gpu/train.sh:34-36). A
// translation unit of its own because it is compiled with other code generation options (build.py: the square
// kernels keep MFMA accumulators in architectural VGPRs; at key width 256 they have to live in AGPRs).
#define DPA_ATTN_WIDE 1
#include "attention.hip"
