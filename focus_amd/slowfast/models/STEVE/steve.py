"""STEVE slot attention over video -- the iterative slot update (mirror of
slowfast/models/STEVE/steve.py:11-105).  The dVAE / CNN encoder / autoregressive decoder around it are
"next" in SURVEY.md section 8(f) and are not built yet."""
import torch
import torch.nn as nn

from focus_amd import ops

from .transformer import TransformerEncoder
from .utils import gru_cell, linear


class SlotAttentionVideo(nn.Module):
    def __init__(self, num_iterations, num_slots, input_size, slot_size, mlp_hidden_size, num_predictor_blocks=1,
                 num_predictor_heads=4, dropout=0.1, epsilon=1e-8):
        super().__init__()
        self.num_iterations = num_iterations
        self.num_slots = num_slots
        self.input_size = input_size
        self.slot_size = slot_size
        self.mlp_hidden_size = mlp_hidden_size
        self.epsilon = epsilon
        self.slot_mu = nn.Parameter(torch.Tensor(1, 1, slot_size))
        self.slot_log_sigma = nn.Parameter(torch.Tensor(1, 1, slot_size))
        nn.init.xavier_uniform_(self.slot_mu)
        nn.init.xavier_uniform_(self.slot_log_sigma)
        self.norm_inputs = nn.LayerNorm(input_size)
        self.norm_slots = nn.LayerNorm(slot_size)
        self.norm_mlp = nn.LayerNorm(slot_size)
        self.project_q = linear(slot_size, slot_size, bias=False)
        self.project_k = linear(input_size, slot_size, bias=False)
        self.project_v = linear(input_size, slot_size, bias=False)
        self.gru = gru_cell(slot_size, slot_size)
        self.mlp = nn.Sequential(linear(slot_size, mlp_hidden_size, weight_init="kaiming"), nn.ReLU(),
                                 linear(mlp_hidden_size, slot_size))
        self.predictor = TransformerEncoder(num_predictor_blocks, slot_size, num_predictor_heads, dropout)

    def forward(self, inputs, noise=None):
        """inputs [B,T,N,Din] -> (slots [B,T,K,Ds], attn_vis [B,T,N,K]).
        `noise` [B,K,Ds] ~ N(0,1): the draw the reference makes inside forward (steve.py:56); drawn here with
        the same call when not supplied, accepted as an argument so parity tests can fix it."""
        B, T, N, Din = inputs.shape
        K, Ds = self.num_slots, self.slot_size
        if noise is None:
            noise = inputs.new_empty(B, K, Ds).normal_()
        slots = (self.slot_mu + torch.exp(self.slot_log_sigma) * noise.float()).to(inputs.dtype)
        ni, ns, nm = self.norm_inputs, self.norm_slots, self.norm_mlp
        k_scale = Ds ** -0.5
        attns_collect, slots_collect = [], []
        frames = ops.unbind_frames(inputs) if inputs.requires_grad else [inputs[:, t] for t in range(T)]
        # every parameter below is applied T x num_iterations times: their gradients are formed once, at the end of
        # the backward pass, from the stacked applications (ops.deferred_wgrads)
        with ops.deferred_wgrads():
            return self._loop(frames, slots, B, T, K, Ds, k_scale)

    def _loop(self, frames, slots, B, T, K, Ds, k_scale):
        ni, ns, nm = self.norm_inputs, self.norm_slots, self.norm_mlp
        attns_collect, slots_collect = [], []
        for t in range(T):
            # LayerNorm + k/v projections of frame t (per frame instead of whole-video: same values, and the
            # per-frame gradients need no zero-padded whole-video buffers)
            x_t = ops.layer_norm(frames[t], ni.weight, ni.bias, ni.eps)
            k_t = ops.linear(x_t, self.project_k.weight, alpha=k_scale)      # k * Ds^-0.5 in the GEMM epilogue
            v_t = ops.linear(x_t, self.project_v.weight)
            for i in range(self.num_iterations):
                slots_prev = slots
                q = ops.linear(ops.layer_norm(slots, ns.weight, ns.bias, ns.eps), self.project_q.weight)
                updates, attn_vis = ops.slot_attn_step(k_t, v_t, q, self.epsilon)          # :76-83
                slots = ops.gru_cell(updates.view(-1, Ds), slots_prev.reshape(-1, Ds), self.gru.weight_ih,
                                     self.gru.weight_hh, self.gru.bias_ih, self.gru.bias_hh).view(B, K, Ds)
                if i < self.num_iterations - 1:
                    y = ops.layer_norm(slots, nm.weight, nm.bias, nm.eps)
                    slots = ops.mlp(y, self.mlp[0].weight, self.mlp[0].bias, self.mlp[2].weight, self.mlp[2].bias,
                                    residual=slots, act=ops.EPI_RELU)
            attns_collect.append(attn_vis)
            slots_collect.append(slots)
            slots = self.predictor(slots)
        return torch.stack(slots_collect, dim=1), torch.stack(attns_collect, dim=1)
