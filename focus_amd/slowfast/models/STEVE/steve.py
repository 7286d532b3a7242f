"""STEVE: slot attention over video with a dVAE token target and an autoregressive transformer decoder (mirror of
slowfast/models/STEVE/steve.py).  The iterative slot update (SlotAttentionVideo, :11-105) is the hot path and runs on
the HIP kernels; the model around it (STEVE.forward, :253-330 -- SURVEY.md section 8(f) rank 1) keeps the reference's
module tree and state_dict keys: convolutions (dVAE, CNN encoder) and the Gumbel-softmax stay on ATen/MIOpen, every
Linear / LayerNorm / FFN / attention of the encoder MLP, the slot projection and the decoder goes through the C ABI."""
import os

import torch
import torch.nn as nn
import torch.nn.functional as F

from focus_amd import ops

from ..build import MODEL_REGISTRY
from .dvae import dVAE
from .transformer import TransformerDecoder, TransformerEncoder
from .utils import Conv2dBlock, conv2d, gru_cell, gumbel_softmax, linear

# Next-frame k / v on a side stream (see _loop_fused): opt-in.  Eager runs are bit-reproducible with it, but whole-step graph
# replays are not (tools/steve_pipeline_check.py, profiles/r03_steve_pipeline_check.txt: first d(k), d(v) reaching the
# projection's backward before they were complete -- fenced since --, and still 9 of 30 replays whose forward diverges at one
# frame: a slot-attention call that read unfinished keys / values).  Off until that is understood.
_PIPELINE_KV = os.environ.get("FOCUS_STEVE_PIPELINE", "0") != "0"
_PIPELINE_JOIN = os.environ.get("FOCUS_STEVE_PIPELINE_JOIN", "event")      # "event" | "stream" (tools/steve_pipeline_check.py)
_PIPELINE_AHEAD1 = os.environ.get("FOCUS_STEVE_PIPELINE_AHEAD1", "0") != "0"
_PIPELINE_FENCE = os.environ.get("FOCUS_STEVE_PIPELINE_FENCE", "0") != "0"
_SIDE_STREAMS = {}


def _side_stream(device):
    s = _SIDE_STREAMS.get(device)
    if s is None:
        s = _SIDE_STREAMS[device] = torch.cuda.Stream(device=device)
    return s


class _KVFence(torch.autograd.Function):
    """Identity on (k_t, v_t) at the point where the main stream takes over the keys and values a side stream produced.
    Forward: the main stream waits for the producer's event.  Backward (it runs on the main stream, after the frame's
    d(k), d(v) have been enqueued there): the side stream is made to wait for the main stream EXPLICITLY before the gradients
    travel on to the projection's backward on the side stream -- the engine's own producer / consumer hand-over alone left
    whole-step graph replays with occasionally stale d(k), d(v) in that product (tools/steve_pipeline_check.py)."""

    @staticmethod
    def forward(ctx, k, v, main, side, ready):
        if _PIPELINE_JOIN == "stream":        # experiment knob: join on everything the side stream has been given so far
            main.wait_stream(side)
        else:
            main.wait_event(ready)
        k.record_stream(main)
        v.record_stream(main)
        ctx.streams = (main, side)
        return k.view_as(k), v.view_as(v)

    @staticmethod
    def backward(ctx, dk, dv):
        main, side = ctx.streams
        side.wait_stream(torch.cuda.current_stream())
        return dk, dv, None, None, None


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
        # every parameter below is applied T x num_iterations times: their gradients are formed once, at the end of
        # the backward pass, from the stacked applications (ops.deferred_wgrads)
        with ops.deferred_wgrads():
            return self._loop(inputs.contiguous(), slots, B, T, K, Ds, k_scale)

    def _loop(self, inputs, slots, B, T, K, Ds, k_scale):
        ni, ns, nm = self.norm_inputs, self.norm_slots, self.norm_mlp
        attns_collect, slots_collect = [], []
        video_grad = ops.FrameGrad()                              # one d(inputs) buffer written by all frames' LN nodes
        tail = ops.SlotTailParams(self.gru, nm, self.mlp, ns, self.project_q)
        fused_tail = ops.slot_tail_ok(slots, tail)                # GRU -> LN -> MLP -> LN -> q as ONE launch per iteration
        if fused_tail:
            return self._loop_fused(inputs, slots, B, T, K, Ds, k_scale, tail, video_grad)
        for t in range(T):
            # LayerNorm + k/v projections of frame t (per frame instead of whole-video: same values; the frame is read
            # in place and its gradient rows are written in place: ops.layer_norm_frame)
            x_t = ops.layer_norm_frame(inputs, t, ni.weight, ni.bias, ni.eps, video_grad)
            # k * Ds^-0.5 in the GEMM epilogue; one node for both projections: d(x_t) = dk.Wk + dv.Wv in one pass
            k_t, v_t = ops.linear_kv(x_t, self.project_k.weight, self.project_v.weight, alpha_k=k_scale)
            kv_grad = ops.SlotKVGrad()                            # d(k_t), d(v_t) of the iterations summed in-kernel
            for i in range(self.num_iterations):
                # layer_norm_fork: the tensor is used twice (normalised, and as GRU state / residual); the gradient of the
                # second use is added inside the LayerNorm backward kernel instead of by an autograd accumulation add
                slots_prev, sn = ops.layer_norm_fork(slots, ns.weight, ns.bias, ns.eps)
                q = ops.linear(sn, self.project_q.weight)
                updates, attn_vis = ops.slot_attn_step(k_t, v_t, q, self.epsilon, kv_grad)  # :76-83
                slots = ops.gru_cell(updates.view(-1, Ds), slots_prev.reshape(-1, Ds), self.gru.weight_ih,
                                     self.gru.weight_hh, self.gru.bias_ih, self.gru.bias_hh).view(B, K, Ds)
                if i < self.num_iterations - 1:
                    sr, y = ops.layer_norm_fork(slots, nm.weight, nm.bias, nm.eps)
                    slots = ops.mlp(y, self.mlp[0].weight, self.mlp[0].bias, self.mlp[2].weight, self.mlp[2].bias,
                                    residual=sr, act=ops.EPI_RELU)
            attns_collect.append(attn_vis)
            slots_collect.append(slots)
            if t < T - 1:
                # (steve.py:100 also runs the predictor after the LAST frame and discards the result: no output and no
                # gradient depends on it, and a dead application would keep the stacked parameter gradients of
                # ops.deferred_wgrads waiting for a backward that never comes)
                slots = self.predictor(slots)
        return torch.stack(slots_collect, dim=1), torch.stack(attns_collect, dim=1)


    def _loop_fused(self, inputs, slots, B, T, K, Ds, k_scale, tail, video_grad):
        """The same loop with the recurrent tail of every iteration in one launch (ops.slot_tail): per frame one "q only"
        call on the incoming slots, then per iteration the slot attention and one tail call (GRU; + residual MLP and the
        next query unless it is the last iteration)."""
        ni = self.norm_inputs
        attns_collect, slots_collect = [], []
        slots = slots.reshape(B * K, Ds)

        def keys_values(t):
            # LayerNorm of frame t and its k / v projection: the only work of a frame that does not depend on the slots
            x_t = ops.layer_norm_frame(inputs, t, ni.weight, ni.bias, ni.eps, video_grad)
            return ops.linear_kv(x_t, self.project_k.weight, self.project_v.weight, alpha_k=k_scale)

        # Frame t + 1's keys and values are produced on a side stream while frame t's iterations run: the recurrence is a
        # chain of narrow launches (the 16-row tails and the predictor keep ~22 of 256 CUs busy) next to which these two
        # full-width kernels -- and, in the backward, the k / v gradient products and the frame LayerNorm's backward, which
        # autograd runs on the stream of their forward -- find idle CUs.  Same values: nothing here depends on the order.
        pipe = _PIPELINE_KV and inputs.is_cuda
        if pipe:
            main, side = torch.cuda.current_stream(), _side_stream(inputs.device)
            if torch.distributed.is_available() and torch.distributed.is_initialized():
                from focus_amd import parallel
                parallel.note_grad_stream(main)
                parallel.note_grad_stream(side)
            side.wait_stream(main)
            inputs.record_stream(side)
            with torch.cuda.stream(side):
                ahead = keys_values(0)
                ready = side.record_event()
        for t in range(T):
            if pipe:
                k_t, v_t = _KVFence.apply(ahead[0], ahead[1], main, side, ready)
                if t + 1 < T:
                    if _PIPELINE_AHEAD1:      # experiment knob: the side stream at most one frame ahead of the recurrence
                        side.wait_stream(main)
                    with torch.cuda.stream(side):
                        ahead = keys_values(t + 1)
                        if _PIPELINE_FENCE:   # experiment knob: a tiny kernel that reads k, v between the products and the join
                            ops.cast(ahead[0].reshape(-1)[:8], torch.float32)
                            ops.cast(ahead[1].reshape(-1)[:8], torch.float32)
                        ready = side.record_event()
            else:
                k_t, v_t = keys_values(t)
            kv_grad = ops.SlotKVGrad()
            slots, q = ops.slot_tail(None, slots, tail, gru=False, mlp=False, q=True)
            for i in range(self.num_iterations):
                updates, attn_vis = ops.slot_attn_step(k_t, v_t, q.view(B, K, Ds), self.epsilon, kv_grad)  # :76-83
                last = i == self.num_iterations - 1
                slots, q = ops.slot_tail(updates.view(-1, Ds), slots, tail, gru=True, mlp=not last, q=not last)
            attns_collect.append(attn_vis)
            slots_collect.append(slots.view(B, K, Ds))
            if t < T - 1:
                slots = self.predictor(slots.view(B, K, Ds)).reshape(B * K, Ds)
        return torch.stack(slots_collect, dim=1), torch.stack(attns_collect, dim=1)


class LearnedPositionalEmbedding1D(nn.Module):
    """steve.py:108-122 (dropout 0.1 on the sum, ATen, training mode only)."""

    def __init__(self, num_inputs, input_size, dropout=0.1):
        super().__init__()
        self.dropout = nn.Dropout(dropout)
        self.pe = nn.Parameter(torch.zeros(1, num_inputs, input_size), requires_grad=True)
        nn.init.trunc_normal_(self.pe)

    def forward(self, input, offset=0):
        T = input.shape[1]
        return self.dropout(input + self.pe[:, offset:offset + T].to(input.dtype))


class CartesianPositionalEmbedding(nn.Module):
    """steve.py:125-145: a 1x1 conv of the (x, y, 1-x, 1-y) grid added to the CNN features."""

    def __init__(self, channels, image_size):
        super().__init__()
        self.projection = conv2d(4, channels, 1)
        self.pe = nn.Parameter(self.build_grid(image_size).unsqueeze(0), requires_grad=False)

    def build_grid(self, side_length):
        coords = torch.linspace(0.0, 1.0, side_length + 1)
        coords = 0.5 * (coords[:-1] + coords[1:])
        grid_y, grid_x = torch.meshgrid(coords, coords, indexing="ij")
        return torch.stack((grid_x, grid_y, 1 - grid_x, 1 - grid_y), dim=0)

    def forward(self, inputs):
        pos = self.projection(self.pe)
        if pos.dtype != inputs.dtype:
            # features from a bf16 convolution stack (TRAIN.MIXED_PRECISION): the sum stays in their type, as the two fp16
            # operands of the reference's autocast do, instead of promoting 600 MB of features to fp32 and back
            pos = pos.to(inputs.dtype)
        return inputs + pos


class OneHotDictionary(nn.Module):
    """steve.py:147-159: embedding of the arg-max token."""

    def __init__(self, vocab_size, emb_size):
        super().__init__()
        self.dictionary = nn.Embedding(vocab_size, emb_size)

    def forward(self, x):
        return self.dictionary(torch.argmax(x, dim=-1))


class BaseCNN(nn.Module):
    """steve.py:162-174."""

    def __init__(self, args):
        super().__init__()
        self.fenc = nn.Sequential(
            Conv2dBlock(args.SLOTS.IMG_CHANNELS, args.SLOTS.CNN_HID_SIZE, 5, 1 if args.SLOTS.IMG_SIZE == 64 else 2, 2),
            Conv2dBlock(args.SLOTS.CNN_HID_SIZE, args.SLOTS.CNN_HID_SIZE, 5, 1, 2),
            Conv2dBlock(args.SLOTS.CNN_HID_SIZE, args.SLOTS.CNN_HID_SIZE, 5, 1, 2),
            conv2d(args.SLOTS.CNN_HID_SIZE, args.SLOTS.DECODER.DIM, 5, 1, 2),
        )

    def forward(self, x):
        return self.fenc(x)


class Res18Block(nn.Module):
    """steve.py:176-203 (needs torchvision's resnet18, which this image does not ship: fails loudly when asked for)."""

    def __init__(self, args):
        super().__init__()
        try:
            from torchvision.models import resnet18
        except ImportError as e:
            raise NotImplementedError("MODEL.CNN_NAME='res18' needs torchvision (steve.py:179)") from e
        self.res18 = resnet18()
        self.res18.conv1 = nn.Conv2d(args.SLOTS.IMG_CHANNELS, args.SLOTS.CNN_HID_SIZE, 3, 1, 1)
        self.fenc = nn.Sequential(*list(self.res18.children())[:-5])
        self.upconv = nn.ConvTranspose2d(args.SLOTS.CNN_HID_SIZE, args.SLOTS.DECODER.DIM, 3, stride=2, padding=1, dilation=1,
                                         output_padding=1)

    def forward(self, x):
        return self.upconv(F.relu(self.fenc(x)))


def fetch_visual_encoder(args):
    if args.MODEL.CNN_NAME == "base":
        return BaseCNN(args)
    if args.MODEL.CNN_NAME == "res18":
        return Res18Block(args)
    raise ValueError("Incorrect cnn name provided!")


class STEVEEncoder(nn.Module):
    """steve.py:215-236."""

    def __init__(self, args):
        super().__init__()
        self.cnn = fetch_visual_encoder(args)
        self.pos = CartesianPositionalEmbedding(args.SLOTS.DECODER.DIM,
                                                args.SLOTS.IMG_SIZE if args.SLOTS.IMG_SIZE == 64 else args.SLOTS.IMG_SIZE // 2)
        self.layer_norm = nn.LayerNorm(args.SLOTS.DECODER.DIM)
        self.mlp = nn.Sequential(linear(args.SLOTS.DECODER.DIM, args.SLOTS.DECODER.DIM, weight_init="kaiming"), nn.ReLU(),
                                 linear(args.SLOTS.DECODER.DIM, args.SLOTS.DECODER.DIM))
        self.savi = SlotAttentionVideo(args.SLOTS.NUM_ITERS, args.SLOTS.NUM_SLOTS, args.SLOTS.DIM, args.SLOTS.SIZE,
                                       args.SLOTS.MLP_HID_SIZE, args.SLOTS.NUM_PREDICTOR_BLOCKS,
                                       args.SLOTS.NUM_PREDICTOR_HEADS, args.SLOTS.PREDICTOR_DROPOUT)
        self.slot_proj = linear(args.SLOTS.SIZE, args.SLOTS.DIM, bias=False)


class STEVEDecoder(nn.Module):
    """steve.py:239-251."""

    def __init__(self, args):
        super().__init__()
        self.dict = OneHotDictionary(args.SLOTS.VOCAB_SIZE, args.SLOTS.DECODER.DIM)
        self.bos = nn.Parameter(torch.Tensor(1, 1, args.SLOTS.DECODER.DIM))
        nn.init.xavier_uniform_(self.bos)
        self.pos = LearnedPositionalEmbedding1D(1 + (args.SLOTS.IMG_SIZE // 4) ** 2, args.SLOTS.DECODER.DIM)
        self.tf = TransformerDecoder(args.SLOTS.DECODER.NUM_BLOCKS, (args.SLOTS.IMG_SIZE // 4) ** 2, args.SLOTS.DECODER.DIM,
                                     args.SLOTS.DECODER.NUM_HEADS, args.SLOTS.DECODER.DROPOUT)
        self.head = linear(args.SLOTS.DECODER.DIM, args.SLOTS.VOCAB_SIZE, bias=False)


@MODEL_REGISTRY.register()
class STEVE(nn.Module):
    """steve.py:253-392.  forward(video [B,T,C,H,W], tau, hard) -> (recon clamped to [0,1], cross_entropy, dvae_mse,
    attns [B,T,K,C,H,W]).  `compute_dtype` (torch.bfloat16 under TRAIN.MIXED_PRECISION) is the storage type of the token
    path (encoder MLP, slot attention, decoder); the convolutions, the Gumbel-softmax and the losses stay fp32.
    `noise` = dict(gumbel_soft, gumbel_hard: Exp(1) draws of the two gumbel_softmax calls; slots: the N(0,1) slot
    initialisation) fixes the random draws for parity tests; each is drawn as the reference draws it when absent."""

    def __init__(self, args):
        super().__init__()
        self.num_iterations = args.SLOTS.NUM_ITERS
        self.num_slots = args.SLOTS.NUM_SLOTS
        self.cnn_hidden_size = args.SLOTS.CNN_HID_SIZE
        self.slot_size = args.SLOTS.SIZE
        self.mlp_hidden_size = args.SLOTS.MLP_HID_SIZE
        self.img_channels = args.SLOTS.IMG_CHANNELS
        self.image_size = args.SLOTS.IMG_SIZE
        self.vocab_size = args.SLOTS.VOCAB_SIZE
        self.d_model = args.SLOTS.DECODER.DIM
        self.compute_dtype = torch.bfloat16 if args.TRAIN.MIXED_PRECISION else torch.float32
        # build-owned key: the slot update (SlotAttentionVideo, ~2000 small launches forward + backward) is captured once
        # into a pair of HIP graphs (torch.cuda.make_graphed_callables) and replayed inside the eager training step
        self.graph_slot_update = bool(args.SLOTS.get("GRAPH_SLOT_UPDATE", False))
        self._savi_graphs = {}
        self.dvae = dVAE(args.SLOTS.VOCAB_SIZE, args.SLOTS.IMG_CHANNELS)
        self.steve_encoder = STEVEEncoder(args)
        self.steve_decoder = STEVEDecoder(args)
        # The convolutions (dVAE, CNN encoder: ATen / MIOpen) run channels-last: MIOpen's kernels for these shapes are NHWC
        # ones and an NCHW caller pays a layout transpose before and after every convolution (batched_transpose_32x32: 6 ms of
        # a 162 ms step); the token path wants [B*T, H*W, C] rows anyway.  Shapes, values and state_dict keys are unchanged.
        self.channels_last = os.environ.get("FOCUS_STEVE_CHANNELS_LAST", "1") != "0"
        # FOCUS_STEVE_CONV_BF16=0: the convolutions in fp32 whatever the compute type; FOCUS_STEVE_ROWS=0: the Gumbel-softmax
        # passes over the vocabulary as ATen launches
        self.conv_dtype = torch.bfloat16 if (self.compute_dtype == torch.bfloat16 and
                                             os.environ.get("FOCUS_STEVE_CONV_BF16", "1") != "0") else None
        self.fused_rows = os.environ.get("FOCUS_STEVE_ROWS", "1") != "0"
        if self.channels_last:
            self.dvae.to(memory_format=torch.channels_last)
            self.steve_encoder.cnn.to(memory_format=torch.channels_last)

    def _conv(self, stack, x):
        """A convolution stack (dVAE decoder, CNN encoder; MIOpen) in the step's compute type: bf16 under
        TRAIN.MIXED_PRECISION -- where the reference's fp16 autocast runs them (steve_train_net.py:95) -- else fp32.  The
        dVAE ENCODER stays fp32 in both modes: its output are the logits of a 4096-way softmax whose gradient is divided by
        tau, bf16 gains 8 % on that stack (the vocabulary projection dominates it) and costs a tenth of the gradient's
        accuracy (tests/test_gpu_steve.py); the sample it feeds the decoder is written in bf16 by the Gumbel kernel."""
        if self.conv_dtype is None or not x.is_cuda:
            return stack(x)
        with torch.autocast("cuda", dtype=self.conv_dtype):
            return stack(x)

    def _frames(self, video):
        """[B,T,C,H,W] -> [B*T,C,H,W] in the memory format the convolutions run in."""
        flat = video.flatten(end_dim=1)
        return flat.contiguous(memory_format=torch.channels_last) if self.channels_last else flat

    # ---- shared by forward / encode (steve.py:294-313, :333-353) ----
    def _slots(self, video, noise=None):
        B, T, C, H, W = video.shape
        enc = self.steve_encoder
        emb = enc.pos(self._conv(enc.cnn, self._frames(video)))                                       # B*T, d_model, H_enc, W_enc
        H_enc, W_enc = emb.shape[-2:]
        emb_set = emb.permute(0, 2, 3, 1).flatten(start_dim=1, end_dim=2).to(self.compute_dtype)    # B*T, H_enc*W_enc, d_model
        ln = enc.layer_norm
        emb_set = ops.mlp(ops.layer_norm(emb_set, ln.weight, ln.bias, ln.eps), enc.mlp[0].weight, enc.mlp[0].bias,
                          enc.mlp[2].weight, enc.mlp[2].bias, act=ops.EPI_RELU)
        emb_set = emb_set.reshape(B, T, H_enc * W_enc, self.d_model)
        if self.graph_slot_update and self.training and emb_set.is_cuda and torch.is_grad_enabled() and emb_set.requires_grad:
            slots, attns = self._savi_graphed(emb_set, noise)
        else:
            slots, attns = enc.savi(emb_set, noise=noise)                                 # [B,T,K,Ds], [B,T,N,K]
        attns = attns.float().transpose(-1, -2).reshape(B, T, self.num_slots, 1, H_enc, W_enc) \
            .repeat_interleave(H // H_enc, dim=-2).repeat_interleave(W // W_enc, dim=-1)  # B, T, K, 1, H, W
        return slots, attns

    def _savi_graphed(self, emb_set, noise):
        """The slot update through a captured forward graph and a captured backward graph (one replay each per step).
        The N(0,1) slot initialisation is drawn OUTSIDE the graph with the call SlotAttentionVideo.forward would make
        (steve.py:56), so the random stream of the step is the eager one.  The graphs read the weights through the bf16
        shadows / stacked operands of focus_amd.ops, which the optimizer's post-step hook refreshes in place."""
        savi = self.steve_encoder.savi
        emb_set = emb_set.contiguous()
        if noise is None:
            noise = emb_set.new_empty(emb_set.shape[0], savi.num_slots, savi.slot_size).normal_()
        key = (tuple(emb_set.shape), emb_set.dtype, noise.dtype)
        g = self._savi_graphs.get(key)
        if g is None:
            state = torch.cuda.get_rng_state(emb_set.device)         # capture + warm-up must not move the random stream
            sample = (torch.zeros_like(emb_set).normal_().requires_grad_(), torch.zeros_like(noise).normal_())
            g = self._savi_graphs[key] = torch.cuda.make_graphed_callables(savi, sample, num_warmup_iters=2)
            torch.cuda.set_rng_state(state, emb_set.device)
        return g(emb_set, noise)

    def forward(self, video, tau, hard, noise=None):
        B, T, C, H, W = video.size()
        noise = noise or {}
        dec = self.steve_decoder
        video_flat = self._frames(video)                                                  # B*T, C, H, W

        # dvae encode (:262-271).  The vocabulary axis is moved LAST: the rows of the channels-last logits are what the row
        # kernels read, and the ATen passes of the other branch run on a contiguous axis instead of as strided dim=1
        # ("spatial") kernels.  Same values; the decoder's 1x1 convolution receives the channels-last view (MIOpen's layout).
        enc_out = self.dvae.encoder(video_flat).permute(0, 2, 3, 1)                       # B*T, H_enc, W_enc, vocab
        last = lambda e: None if e is None else e.permute(0, 2, 3, 1)
        e_soft, e_hard = last(noise.get("gumbel_soft")), last(noise.get("gumbel_hard"))
        rows = enc_out.reshape(-1, self.vocab_size)                                       # a view of channels-last logits
        if self.fused_rows and (e_soft is None) == (e_hard is None) and ops.rows_ok(rows):
            # log_softmax, both Gumbel perturbations, the relaxed sample and the hard sample's arg-max in one pass over the
            # rows (csrc/gumbel.hip); noise not given = drawn in the kernel
            flat = lambda e: None if e is None else e.reshape(-1, self.vocab_size)
            z_rows, target = ops.gumbel_softmax_rows(rows, tau, hard, flat(e_soft), flat(e_hard),
                                                     out_dtype=self.conv_dtype)
            z_soft = z_rows.view(enc_out.shape).permute(0, 3, 1, 2)
            target = target.view(enc_out.shape[0], -1)
        else:
            z_logits = F.log_softmax(enc_out.float(), dim=-1)
            z_soft = gumbel_softmax(z_logits, tau, hard, dim=-1, noise=e_soft).permute(0, 3, 1, 2)
            # the hard sample is only ever read through its arg-max (:268-269), which is the arg-max of the perturbed logits:
            # softmax is monotonic, so no second softmax, one-hot scatter and straight-through sum over 4096 x 1024 x B*T values
            if e_hard is None:
                e_hard = torch.empty_like(z_logits).exponential_()
            with torch.no_grad():
                target = (z_logits - (e_hard + torch.finfo(z_logits.dtype).tiny).log()).argmax(dim=-1).flatten(start_dim=1)
        z_emb = dec.dict.dictionary(target)                                               # B*T, H_enc*W_enc, d_model
        z_emb = torch.cat([dec.bos.expand(B * T, -1, -1), z_emb], dim=1)
        z_emb = dec.pos(z_emb)

        # dvae recon (:274-275)
        # (the squared error is summed per frame, then over frames: ATen's single-pass reduction of all 9 M elements to one
        # scalar takes 1.3 ms here; rows are taken in the memory order of the convolutions' layout)
        recon_flat = self._conv(self.dvae.decoder, z_soft).float()                        # B*T, C, H, W
        err = video_flat - recon_flat
        err = err.permute(0, 2, 3, 1) if self.channels_last else err
        dvae_mse = err.reshape(B * T, -1).square().sum(dim=1).sum() / (B * T)
        dvae_recon = recon_flat.reshape(B, T, C, H, W)

        # slots (:277-300)
        slots, attns = self._slots(video, noise.get("slots"))
        attns = video.unsqueeze(2) * attns + (1.0 - attns)                                # B, T, K, C, H, W

        # decode (:303-306): cross entropy of the hard tokens under the decoder's prediction, summed over the tokens
        slots = ops.linear(slots, self.steve_encoder.slot_proj.weight)                    # B, T, K, d_model
        pred = dec.tf(z_emb[:, :-1].to(self.compute_dtype), slots.flatten(end_dim=1))     # B*T, H_enc*W_enc, d_model
        pred = ops.linear(pred, dec.head.weight)                                          # B*T, H_enc*W_enc, vocab
        ntok = pred.shape[1]
        cross_entropy = ops.label_smoothing_ce(pred.reshape(-1, self.vocab_size), target.reshape(-1), 0.0) * ntok
        return dvae_recon.clamp(0.0, 1.0), cross_entropy, dvae_mse, attns

    def encode(self, video):
        """steve.py:331-357 -> (slots, attns_vis, attns)."""
        slots, attns = self._slots(video)
        attns_vis = video.unsqueeze(2) * attns + (1.0 - attns)
        return slots, attns_vis, attns

    def decode(self, slots):
        """steve.py:359-381: greedy autoregressive token generation, then the dVAE decoder."""
        B, num_slots, slot_size = slots.size()
        H_enc, W_enc = self.image_size // 4, self.image_size // 4
        gen_len = H_enc * W_enc
        dec = self.steve_decoder
        slots = ops.linear(slots.to(self.compute_dtype), self.steve_encoder.slot_proj.weight)
        z_gen = slots.new_zeros(0, dtype=torch.long)
        input = dec.bos.expand(B, 1, -1)
        for _ in range(gen_len):
            decoder_output = dec.tf(dec.pos(input).to(self.compute_dtype), slots)
            z_next = ops.linear(decoder_output[:, -1:].contiguous(), dec.head.weight).argmax(dim=-1)      # B, 1
            z_gen = torch.cat((z_gen, z_next), dim=1)
            input = torch.cat((input, dec.dict.dictionary(z_next)), dim=1)
        z_gen = F.one_hot(z_gen, self.vocab_size).transpose(1, 2).float().reshape(B, -1, H_enc, W_enc)
        return self.dvae.decoder(z_gen).clamp(0.0, 1.0)

    def reconstruct_autoregressive(self, video):
        """steve.py:383-392."""
        B, T, C, H, W = video.size()
        slots, attns, _ = self.encode(video)
        return self.decode(slots.flatten(end_dim=1)).reshape(B, T, C, H, W)
