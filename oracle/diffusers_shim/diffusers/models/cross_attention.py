"""CrossAttention + default processor (diffusers 0.13.1 `models/cross_attention.py` semantics).

Processor maths: q=to_q(h); k,v=to_k/to_v(ctx or h); heads folded into batch; scores = baddbmm(alpha=scale)
-> softmax(dim=-1) -> bmm -> heads merged -> to_out[0] (Linear+bias) -> to_out[1] (Dropout).
The score matrix is evaluated in query chunks purely to bound host memory ([24,4096,12288] fp32 = 4.8 GB
otherwise); softmax is row-wise so the result is identical.
"""
import torch
from torch import nn


class CrossAttnProcessor:
    query_chunk = 1024

    def __call__(self, attn, hidden_states, encoder_hidden_states=None, attention_mask=None):
        query = attn.to_q(hidden_states)
        ctx = hidden_states if encoder_hidden_states is None else encoder_hidden_states
        key = attn.to_k(ctx)
        value = attn.to_v(ctx)
        query = attn.head_to_batch_dim(query)
        key = attn.head_to_batch_dim(key)
        value = attn.head_to_batch_dim(value)
        outs = []
        for s in range(0, query.shape[1], self.query_chunk):
            q = query[:, s:s + self.query_chunk]
            probs = attn.get_attention_scores(q, key, attention_mask)
            outs.append(torch.bmm(probs, value))
        hidden_states = torch.cat(outs, dim=1)
        hidden_states = attn.batch_to_head_dim(hidden_states)
        hidden_states = attn.to_out[0](hidden_states)
        hidden_states = attn.to_out[1](hidden_states)
        return hidden_states


class CrossAttention(nn.Module):
    def __init__(self, query_dim, cross_attention_dim=None, heads=8, dim_head=64, dropout=0.0, bias=False,
                 upcast_attention=False, upcast_softmax=False, added_kv_proj_dim=None, norm_num_groups=None,
                 processor=None):
        super().__init__()
        inner_dim = dim_head * heads
        cross_attention_dim = cross_attention_dim if cross_attention_dim is not None else query_dim
        self.upcast_attention = upcast_attention
        self.upcast_softmax = upcast_softmax
        self.scale = dim_head ** -0.5
        self.heads = heads
        self.sliceable_head_dim = heads
        self.to_q = nn.Linear(query_dim, inner_dim, bias=bias)
        self.to_k = nn.Linear(cross_attention_dim, inner_dim, bias=bias)
        self.to_v = nn.Linear(cross_attention_dim, inner_dim, bias=bias)
        self.to_out = nn.ModuleList([nn.Linear(inner_dim, query_dim), nn.Dropout(dropout)])
        self.processor = processor if processor is not None else CrossAttnProcessor()

    def set_use_memory_efficient_attention_xformers(self, valid, attention_op=None):
        pass

    def set_attention_slice(self, slice_size):
        if slice_size is not None and slice_size > self.sliceable_head_dim:
            raise ValueError(f"slice_size {slice_size} has to be smaller or equal to {self.sliceable_head_dim}.")

    def set_processor(self, processor):
        self.processor = processor

    def forward(self, hidden_states, encoder_hidden_states=None, attention_mask=None, **kw):
        return self.processor(self, hidden_states, encoder_hidden_states=encoder_hidden_states,
                              attention_mask=attention_mask, **kw)

    def head_to_batch_dim(self, t):
        b, n, c = t.shape
        h = self.heads
        return t.reshape(b, n, h, c // h).permute(0, 2, 1, 3).reshape(b * h, n, c // h)

    def batch_to_head_dim(self, t):
        bh, n, d = t.shape
        h = self.heads
        return t.reshape(bh // h, h, n, d).permute(0, 2, 1, 3).reshape(bh // h, n, d * h)

    def get_attention_scores(self, query, key, attention_mask=None):
        dtype = query.dtype
        if self.upcast_attention:
            query, key = query.float(), key.float()
        empty = torch.empty(query.shape[0], query.shape[1], key.shape[1], dtype=query.dtype, device=query.device)
        scores = torch.baddbmm(empty, query, key.transpose(-1, -2), beta=0, alpha=self.scale)
        if attention_mask is not None:
            scores = scores + attention_mask
        if self.upcast_softmax:
            scores = scores.float()
        return scores.softmax(dim=-1).to(dtype)
