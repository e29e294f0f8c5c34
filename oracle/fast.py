"""Exact-shortcut variant of the oracle + emulation of the device precisions.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

``FastModel`` computes the same function as oracle/predict.py but organised the
way the HIP path is (DESIGN.md "Algebraic shortcuts"), so that a full-size
1000-step segment is affordable on the CPU and so that the shortcuts themselves
are proven against the faithful restatement (tests/test_oracle_fast.py):

  S1  everything that depends only on the step index (log-SNRs, the five sampler
      coefficients, the time embedding MLP and every FiLM scale/bias) is
      tabulated once per model (diffusion_utils.py:166-187,120-163;
      network.py:377-392; layers.py:660-665).
  S2  cross-attention K/V projections of the encodings are computed once per
      segment, not once per step (network.py:217-230 recomputes them per call).
  S3  padded encoder positions are dropped: a masked key has weight
      exp(-1e10 + s - max) == 0 exactly in fp32 (layers.py:341-346) and padded
      query rows are never read downstream.
  S4  the unconditional CFG pass multiplies encodings AND masks by 0
      (models.py:376-377) -> every key masked -> zero_activations_if_masked
      returns exactly 0 (layers.py:882-902) -> that pass skips cross-attention.
      The same rule zeroes cross-attention when no key at all is valid.

``precision``:
  'f32'     plain arithmetic in the backend dtype (the parity reference)
  'bf16'    GEMM/attention operands rounded to bfloat16 (fp32 accumulate),
            residual stream / norms / softmax / sampler in fp32 -- emulates the
            device 'bf16' mode
  'bf16x3'  operands split hi+lo bf16, three products (error ~2^-16) -- emulates
            the device 'bf16x3' mode
  'f16x3'   the same with IEEE-half planes (22 significand bits; weights times 2^9, as the
            device packs them) -- emulates the device 'f16x3' mode, the default (DESIGN.md 3:
            float32-class at the cost of bf16x3); 'f16' = one half plane
"""
from __future__ import annotations

import math

import numpy as np

from oracle import ops
from oracle import predict as predict_lib
from oracle import sampler as du


class FastModel:

  def __init__(self, xp, cfg, diffusion_config, params, context, precision='f32',
               codec=None):
    assert cfg.decoder_cross_attend_style in ('concat_encodings', 'sum_cross_attends')
    self.sum_style = cfg.decoder_cross_attend_style == 'sum_cross_attends'
    assert diffusion_config.model_output in ('eps', 'x0', 'v')
    assert diffusion_config.sampler.name in ('ddpm', 'ddim')
    self.xp, self.cfg, self.dc = xp, cfg, diffusion_config
    self.context = context
    self.precision = precision
    self.codec = codec or predict_lib.MelGANCodec()
    self.p = {k: xp.asarray(v) for k, v in params.items()}
    self._wcache = {}
    self.tok = 'token_encoder' if context else 'encoder'
    self._build_tables()

  # -- precision emulation ---------------------------------------------------
  def _round16(self, a):
    """One operand plane: bfloat16 (8 significand bits) or, for the 'f16*' modes, IEEE half (11 bits,
    saturating at 65504 as the device conversion would)."""
    return self.xp.round_f16(a) if self.precision.startswith('f16') else self.xp.round_bf16(a)

  def _split(self, a):
    if self.precision == 'f32':
      return (a,)
    hi = self._round16(a)
    if self.precision in ('bf16', 'f16'):
      return (hi,)
    return (hi, self._round16(a - hi))

  W_SCALE_F16 = 512.0   # csrc/common.h kWScale: half-plane weights are packed times 2^9

  def _w(self, name):
    """Weight planes; half planes are taken of the weight times 2^9, as the device packs them, so that
    the lo plane of |w| ~ 0.03 weights stays a normal half; `mm` undoes the scale on the fp32 result,
    exactly."""
    if name not in self._wcache:
      w = self.p[name]
      sc = self.W_SCALE_F16 if self.precision.startswith('f16') else 1.0
      self._wcache[name] = (self._split(w * sc) if sc != 1.0 else self._split(w), sc)
    return self._wcache[name]

  def _mm_parts(self, a_parts, w_parts):
    xp = self.xp
    y = xp.matmul(a_parts[0], w_parts[0])
    if len(a_parts) == 2:
      y = y + xp.matmul(a_parts[0], w_parts[1]) + xp.matmul(a_parts[1], w_parts[0])
    return y

  def mm(self, a, name):
    """Device GEMM: operands in the emulated precision, fp32 accumulate."""
    w_parts, sc = self._w(name)
    y = self._mm_parts(self._split(a), w_parts)
    return y if sc == 1.0 else y * (1.0 / sc)

  def rq(self, a):
    """Storage rounding of a GEMM/attention output that the device keeps in ONE 16-bit plane."""
    if self.precision in ('bf16', 'f16'):
      return self._round16(a)
    return a  # f32; the x3 modes keep hi+lo planes: ~fp32

  # -- S1: step-indexed tables (always full precision) -----------------------
  def _build_tables(self):
    xp, cfg, dc, p = self.xp, self.cfg, self.dc, self.p
    n = dc.sampler.schedule.num_steps
    i = xp.arange(n)
    t = (i + 1.0) / float(n)
    s = i / float(n)
    self.logsnr_t = du.get_logsnr_t(xp, t, dc.sampler.schedule)
    self.logsnr_s = du.get_logsnr_t(xp, s, dc.sampler.schedule)
    # network.py:377-392 for every step at once
    emb = du.get_timing_signal_1d(xp, t * cfg.max_decoder_noise_time, cfg.emb_dim,
                                  max_timescale=cfg.max_decoder_noise_time)
    emb = ops.swish(xp, xp.matmul(emb, p['decoder/time_emb_dense0/kernel']))
    emb = ops.swish(xp, xp.matmul(emb, p['decoder/time_emb_dense1/kernel']))
    self.film = []
    for l in range(cfg.num_decoder_layers):
      pair = []
      for k in range(2):
        sb = xp.matmul(emb, p['decoder/layers_%d/FiLMLayer_%d/DenseGeneral_0/kernel' % (l, k)])
        pair.append(sb)  # [N, 2D]: scale | bias
      self.film.append(pair)

  # -- attention core ---------------------------------------------------------
  def _attend(self, q, k, v):
    """q [rows,H,d], k/v [keys,H,d] -> [rows,H*d]; online-softmax arithmetic of
    the device kernel: P = exp(s - max) rounded for the PV product, row sum in
    fp32, normalisation after PV."""
    xp = self.xp
    s = xp.einsum('qhd,khd->hqk', q, k)
    m = xp.max(s, axis=-1, keepdims=True)
    pr = xp.exp(s - m)
    l = xp.sum(pr, axis=-1, keepdims=True)
    o = xp.einsum('hqk,khd->hqd', self.rq(pr), v) / l
    o = xp.einsum('hqd->qhd', o)
    return xp.reshape(o, (q.shape[0], q.shape[1] * q.shape[2]))

  def _heads(self, x):
    c = self.cfg
    return self.xp.reshape(x, (x.shape[0], c.num_heads, c.head_dim))

  def _self_attention(self, prefix, h):
    q = self.rq(self.mm(h, prefix + '/query/kernel'))
    k = self.rq(self.mm(h, prefix + '/key/kernel'))
    v = self.rq(self.mm(h, prefix + '/value/kernel'))
    a = self.rq(self._attend(self._heads(q), self._heads(k), self._heads(v)))
    return self.mm(a, prefix + '/out/kernel')

  def _mlp(self, prefix, h):
    xp = self.xp
    g = ops.gelu_tanh(xp, self.mm(h, prefix + '/wi_0/kernel')) * self.mm(h, prefix + '/wi_1/kernel')
    return self.mm(self.rq(g), prefix + '/wo/kernel')

  # -- S2/S3: encoders on the valid positions only ----------------------------
  def _encoder_stack(self, prefix, x):
    xp, p, cfg = self.xp, self.p, self.cfg
    for l in range(cfg.num_encoder_layers):
      lp = '%s/layers_%d' % (prefix, l)
      h = ops.rms_layer_norm(xp, x, p[lp + '/pre_attention_layer_norm/scale'])
      x = x + self._self_attention(lp + '/attention', h)
      h = ops.rms_layer_norm(xp, x, p[lp + '/pre_mlp_layer_norm/scale'])
      x = x + self._mlp(lp + '/mlp', h)
    return ops.rms_layer_norm(xp, x, p[prefix + '/encoder_norm/scale'])

  def encode(self, tokens, ctx=None, ctx_mask=None):
    """tokens int [B,L]; ctx [B,C,n] in mel units; ctx_mask int [B,C]."""
    xp, p, cfg = self.xp, self.p, self.cfg
    tokens = np.asarray(tokens)
    self.kv = []
    for b in range(tokens.shape[0]):
      tok_enc = ctx_enc = None
      valid = np.nonzero(tokens[b] > 0)[0]
      if valid.size:
        x = xp.take(p[self.tok + '/token_embedder/embedding'], xp.asint(tokens[b][valid]))
        x = x + xp.take(p[self.tok + '/Embed_0/embedding'], xp.asint(valid))
        tok_enc = self._encoder_stack(self.tok, x)
      if self.context:
        cm = np.asarray(ctx_mask[b])
        cvalid = np.nonzero(cm > 0)[0]
        if cvalid.size:
          c = self.codec.scale_features(xp, xp.asarray(ctx[b]), (-1., 1.), clip=True)
          # input_proj runs in full precision on the device (K = 128, fp32 MFMA)
          x = xp.matmul(c, p['continuous_encoder/input_proj/kernel'])
          if cfg.context_positions == 'terminal_relative':
            zeros = (cm == 0)
            seq_len = int(zeros.argmax()) if zeros.any() else 0
            if seq_len == 0 and cm[0] != 0:
              seq_len = cm.shape[0]
            pos = np.roll(np.arange(cm.shape[0]), seq_len)
          else:
            pos = np.arange(cm.shape[0])
          x = x + xp.take(p['continuous_encoder/Embed_0/embedding'], xp.asint(pos))
          x = xp.take(x, xp.asint(cvalid))
          ctx_enc = self._encoder_stack('continuous_encoder', x)
      # key regions, one cross-attention module each: concat_encodings = ONE region holding both encodings
      # (network.py:217-235); sum_cross_attends = a module per encoding, outputs summed (:199-216)
      if self.sum_style:
        regions = [(0, tok_enc), (1, ctx_enc)] if self.context else [(0, tok_enc)]
      else:
        both = [e for e in (tok_enc, ctx_enc) if e is not None]
        regions = [(0, xp.concatenate(both, 0) if both else None)]
      mods = []
      for e, enc in regions:
        if enc is None:
          continue   # no valid key: zero_activations_if_masked makes the module's output exactly 0
        layers = []
        for l in range(cfg.num_decoder_layers):
          ap = 'decoder/layers_%d/MultiHeadDotProductAttention_%d' % (l, e)
          layers.append((self._heads(self.rq(self.mm(enc, ap + '/key/kernel'))),
                         self._heads(self.rq(self.mm(enc, ap + '/value/kernel')))))
        mods.append((e, layers))
      self.kv.append(mods or None)

  # -- one decoder pass --------------------------------------------------------
  def decoder_pass(self, z, i, cond):
    """z [B,T,n], scan index i -> eps prediction [B,T,n] (network.py:360-457)."""
    xp, p, cfg = self.xp, self.p, self.cfg
    outs = []
    for b in range(z.shape[0]):
      # continuous_inputs_projection + positions: full precision (K = 128)
      x = xp.matmul(z[b], p['decoder/continuous_inputs_projection/kernel'])
      x = x + p['decoder/Embed_0/embedding'][:z.shape[1]]
      d = cfg.emb_dim
      for l in range(cfg.num_decoder_layers):
        lp = 'decoder/layers_%d' % l
        sb = self.film[l][0][i]
        h = ops.rms_layer_norm(xp, x, p[lp + '/pre_self_attention_layer_norm/scale'])
        h = h * (sb[:d] + 1.0) + sb[d:]
        x = x + self._self_attention(lp + '/self_attention', h)
        if cond and self.kv[b] is not None:  # S4
          h = ops.rms_layer_norm(xp, x, p[lp + '/pre_cross_attention_layer_norm/scale'])
          upd = None
          for e, layers in self.kv[b]:
            ap = lp + '/MultiHeadDotProductAttention_%d' % e
            q = self._heads(self.rq(self.mm(h, ap + '/query/kernel')))
            k, v = layers[l]
            o = self.mm(self.rq(self._attend(q, k, v)), ap + '/out/kernel')
            upd = o if upd is None else upd + o
          x = x + upd
        sb = self.film[l][1][i]
        h = ops.rms_layer_norm(xp, x, p[lp + '/pre_mlp_layer_norm/scale'])
        h = h * (sb[:d] + 1.0) + sb[d:]
        x = x + self._mlp(lp + '/mlp', h)
      y = ops.rms_layer_norm(xp, x, p['decoder/decoder_norm/scale'])
      # spec_out_dense is float32 in the reference (network.py:454) and on the device
      outs.append(xp.expand_dims(xp.matmul(y, p['decoder/spec_out_dense/kernel']), 0))
    return xp.concatenate(outs, 0)

  # -- sampler (diffusion_utils.py:398-476) ------------------------------------
  def sample(self, init_z, noise, trace=None):
    xp, dc = self.xp, self.dc
    z = xp.asarray(init_z)
    noise = None if noise is None else xp.asarray(noise)
    n = dc.sampler.schedule.num_steps
    w = dc.classifier_free_guidance.eval_condition_weight
    for i in reversed(range(n)):
      lt = xp.reshape(self.logsnr_t[i], (1,))
      ls = xp.reshape(self.logsnr_s[i], (1,))
      # the model output is converted at the TRAIN schedule's log-SNR (diffusion_utils.py:294)
      tm = xp.full((1,), 0.0) + (float(i) + 1.0)
      tm = tm / float(n)
      o = du.get_x0_and_eps_from_model_output(xp, z, tm, self.decoder_pass(z, i, True), dc)
      eps, x0 = o['eps'], o['x0']
      if w != 1:
        ou = du.get_x0_and_eps_from_model_output(xp, z, tm, self.decoder_pass(z, i, False), dc)
        eps = w * eps + (1. - w) * ou['eps']
        x0 = du.predict_x0_from_eps(xp, z=z, eps=eps, logsnr=lt)
      if dc.sampler.clip_x0:
        x0 = xp.clip(x0, -1.0, 1.0)
        eps = du.predict_eps_from_x0(xp, z=z, x0=x0, logsnr=lt)
      if dc.sampler.name == 'ddim':
        z = du.ddim_step(xp, i, ls, lt, x0, eps)
      else:
        z = du.ddpm_step(xp, i, None if noise is None else noise[i], ls, lt, x0, z,
                         dc.sampler.logvar_type)
      if trace is not None:
        trace.append(z)
    return z

  def predict(self, batch, init_z, noise, trace=None):
    """Same contract as oracle.predict.predict_batch_with_aux."""
    xp = self.xp
    if self.context:
      self.encode(batch['encoder_input_tokens'], batch['encoder_continuous_inputs'],
                  batch['encoder_continuous_mask'])
    else:
      self.encode(batch['encoder_input_tokens'])
    x0 = self.sample(init_z, noise, trace=trace)
    decodes = self.codec.scale_to_features(xp, x0, input_range=(-1., 1.))
    return decodes, xp.zeros((np.asarray(batch['encoder_input_tokens']).shape[0],))
