"""Operator layer: autograd wrappers over the gfx950 HIP kernels, each with an eager PyTorch
implementation that is used only for CPU tensors (plumbing tests) or unsupported shapes."""
from .activation import gelu, get_activation, swiglu
from .attention import (cached_attention, causal_attention, causal_attention_eager, causal_self_attention_qkv, dropout_keep_mask,
                        kv_append, qkv_rope_attention, qkv_rope_attention_supported, split_qkv)
from .decode import ACT_GELU, ACT_NONE, ACT_SWIGLU, decode_linear, decode_linear_eager, decode_linear_supported
from .embedding import embedding
from .linear import (bump_weights_epoch, linear, mlp_gelu, mlp_gelu_supported, mlp_swiglu, mlp_swiglu_supported,
                     transposed_weight, weight_grad,
                     wgrad_supported)
from .loss import (IGNORE_INDEX, cross_entropy, linear_cross_entropy, vocab_parallel_cross_entropy,
                   vocab_parallel_linear_cross_entropy)
from .moe import (expert_counts, gather_combine, grouped_experts, grouped_experts_eager, grouped_layout_block,
                  grouped_supported,
                  plan_dense, scatter_rows, topk_gate)
from .moe import plan as moe_plan
from .norm import add_layer_norm, layer_norm, layer_norm_keep
from .optim import adamw_step, sumsq
from .rope import apply_rope, rope_inplace_, rope_tables, rope_qkv_
from .sampling import sample_eager, sample_fused, sample_step, sample_step_supported

__all__ = [
    "gelu", "swiglu", "get_activation", "cached_attention", "causal_attention", "causal_attention_eager", "causal_self_attention_qkv", "dropout_keep_mask", "kv_append",
    "decode_linear", "decode_linear_eager", "decode_linear_supported", "ACT_NONE", "ACT_GELU", "ACT_SWIGLU",
    "rope_qkv_", "split_qkv", "qkv_rope_attention", "qkv_rope_attention_supported", "embedding", "cross_entropy", "linear_cross_entropy", "vocab_parallel_cross_entropy", "vocab_parallel_linear_cross_entropy",
    "IGNORE_INDEX", "layer_norm", "add_layer_norm", "layer_norm_keep", "adamw_step", "sumsq", "apply_rope", "rope_inplace_",
    "rope_tables", "linear", "mlp_gelu", "mlp_gelu_supported", "mlp_swiglu", "mlp_swiglu_supported", "bump_weights_epoch", "transposed_weight", "weight_grad", "wgrad_supported", "grouped_experts", "grouped_experts_eager",
    "grouped_supported", "expert_counts", "sample_eager", "sample_fused", "sample_step", "sample_step_supported", "topk_gate", "moe_plan", "plan_dense", "scatter_rows", "gather_combine",
]
