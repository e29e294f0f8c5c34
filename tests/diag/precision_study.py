"""CPU studies of the operand arithmetic with the oracle's precision emulation (test infrastructure): the `small`
1000-step golden segment (or, `base`, the two chained base_with_context segments) is re-run with a variant of the
device arithmetic and the mel rms against the float64 fixture is printed next to the north-star bar (1e-3).  The
emulation has predicted the device to three digits every time it was checked (bf16x3 1.40e-4 / 1.376e-4 measured,
f16x3 6.77e-5 / 6.82e-5, base segment 0 6.28e-5 / 6.28e-5).  Guides kernel work; changes nothing in the product.

  x3, pv_*, qk_*, q_p_single   bfloat16 planes: which of the three hi / lo products attention needs
  mm_no_alo, mm_no_wlo         bfloat16 planes: GEMM with one activation / weight plane
  lo8_*                        bfloat16 hi + fp8 lo planes
  f16x3, f16x3_noscale, f16x4, f16x2_no_alo      IEEE-half planes (what the device computes in now)
  f16_noalo_*                  half planes, activation lo plane dropped for one projection class
  f16_noqlo, f16_noplo, f16_noqplo, f16_noqplo_self, f16_noqplo_cross
                               half planes, the QUERY-side lo planes of attention dropped (Q in S = Q.K^T, P in
                               O = P.V): 2 MFMAs per product instead of 3 and no split of P (VERDICT r02 item 5;
                               the bfloat16-plane rows qk_no_qlo / pv_no_plo above predate the half planes)

  python -m tests.diag.precision_study [base] [variant ...]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import msd_amd
from oracle import backend, fast, philox
from tests import helpers
from tests.test_golden import GOLD

VARIANTS = {
    # name: (QK products, PV products); each product = (left part, right part), parts 0 = hi, 1 = lo
    'x3':       ([(0, 0), (0, 1), (1, 0)], [(0, 0), (0, 1), (1, 0)]),   # the device's bf16x3
    'pv_no_plo': ([(0, 0), (0, 1), (1, 0)], [(0, 0), (0, 1)]),          # P single plane
    'pv_no_vlo': ([(0, 0), (0, 1), (1, 0)], [(0, 0), (1, 0)]),          # V single plane
    'pv_1':     ([(0, 0), (0, 1), (1, 0)], [(0, 0)]),
    'qk_no_qlo': ([(0, 0), (0, 1)], [(0, 0), (0, 1), (1, 0)]),          # Q single plane
    'qk_no_klo': ([(0, 0), (1, 0)], [(0, 0), (0, 1), (1, 0)]),          # K single plane
    'qk_1':     ([(0, 0)], [(0, 0), (0, 1), (1, 0)]),
    'q_p_single': ([(0, 0), (0, 1)], [(0, 0), (0, 1)]),                  # Q and P single plane (4 of 6 products)
}


# GEMM variants: products of (activation part, weight part)
MM_VARIANTS = {
    'mm_no_alo': [(0, 0), (0, 1)],     # activations single plane (weights hi + lo)
    'mm_no_wlo': [(0, 0), (1, 0)],     # weights single plane (activations hi + lo)
}


# 8-bit lo planes: the lo plane of the GEMM operands stored as fp8 with one power-of-two
# scale per row of K (what a 16+8-bit operand with v_cvt_scalef32_pk_bf16_fp8 on the way to the MFMA would
# hold): cuts the L2 -> LDS bytes of that operand by 25 %.  'a' = activations, 'w' = weights.
LO8_VARIANTS = {
    'lo8_a_e4m3': ('a', 'e4m3'), 'lo8_w_e4m3': ('w', 'e4m3'), 'lo8_aw_e4m3': ('aw', 'e4m3'),
    'lo8_a_e5m2': ('a', 'e5m2'), 'lo8_w_e5m2': ('w', 'e5m2'),
}


def _fp8_round(xp, lo, kind):
  """lo -> fp8 (per-row power-of-two scale so that the row maximum lands in the top binade) -> back."""
  import torch
  t = lo if isinstance(lo, torch.Tensor) else torch.as_tensor(np.asarray(lo))
  dt = torch.float8_e4m3fn if kind == 'e4m3' else torch.float8_e5m2
  top = 448.0 if kind == 'e4m3' else 57344.0
  amax = t.abs().amax(dim=-1, keepdim=True).clamp_min(1e-30)
  scale = torch.exp2(torch.ceil(torch.log2(amax / top)))
  q = (t / scale).to(dt).to(t.dtype) * scale
  return q


# fp16 planes instead of bf16 planes (same bytes, same MFMA rate: v_mfma_f32_16x16x32_f16): hi + lo carry 22
# significand bits against 16.  Weights are pre-scaled per tensor by a power of two so that lo stays a NORMAL
# fp16 number (|w| ~ 0.03: lo ~ 1e-5 would be subnormal); activations are split as they are, saturating.
F16_VARIANTS = {
    'f16x3': dict(scale_w=True, mm=[(0, 0), (0, 1), (1, 0)]),
    'f16_noalo_mlp_out': dict(scale_w=True, mm=[(0, 0), (0, 1), (1, 0)]),
    'f16_noalo_attn_out': dict(scale_w=True, mm=[(0, 0), (0, 1), (1, 0)]),
    'f16_noalo_mlp_in': dict(scale_w=True, mm=[(0, 0), (0, 1), (1, 0)]),
    'f16_noalo_qkv': dict(scale_w=True, mm=[(0, 0), (0, 1), (1, 0)]),
    'f16_noalo_outs': dict(scale_w=True, mm=[(0, 0), (0, 1), (1, 0)]),
    'f16x3_noscale': dict(scale_w=False, mm=[(0, 0), (0, 1), (1, 0)]),
    'f16x2_no_alo': dict(scale_w=True, mm=[(0, 0), (0, 1)]),      # activations single fp16 plane: 2 MFMAs per product
    'f16x4': dict(scale_w=True, mm=[(0, 0), (0, 1), (1, 0), (1, 1)]),
    'f16_noqlo': dict(scale_w=True, mm=[(0, 0), (0, 1), (1, 0)]),
    'f16_noplo': dict(scale_w=True, mm=[(0, 0), (0, 1), (1, 0)]),
    'f16_noqplo': dict(scale_w=True, mm=[(0, 0), (0, 1), (1, 0)]),
    'f16_noqplo_self': dict(scale_w=True, mm=[(0, 0), (0, 1), (1, 0)]),
    'f16_noqplo_cross': dict(scale_w=True, mm=[(0, 0), (0, 1), (1, 0)]),
    'f16_noqplo_dec': dict(scale_w=True, mm=[(0, 0), (0, 1), (1, 0)]),
    'f16_noqlo_dec': dict(scale_w=True, mm=[(0, 0), (0, 1), (1, 0)]),
}

# half planes, query-side lo planes of ATTENTION dropped: name -> (QK products, PV products, where)
_X3 = [(0, 0), (0, 1), (1, 0)]
_NOL = [(0, 0), (0, 1)]          # left operand (Q resp. P) single plane
F16_ATT = {
    'f16_noqlo': (_NOL, _X3, 'all'), 'f16_noplo': (_X3, _NOL, 'all'), 'f16_noqplo': (_NOL, _NOL, 'all'),
    'f16_noqplo_self': (_NOL, _NOL, 'self'), 'f16_noqplo_cross': (_NOL, _NOL, 'cross'),
    # what the device does since round 3 (attention.h QP = 3 on the DECODER's two attentions; encoders keep all planes)
    'f16_noqplo_dec': (_NOL, _NOL, 'dec'), 'f16_noqlo_dec': (_NOL, _X3, 'dec'),
}


# half planes, activation lo plane dropped for SOME GEMMs only (2 MFMAs per product there and a third less
# activation ingest): which projections tolerate it?  name -> substrings of the weight names affected
F16_ALO_DROP = {
    'f16_noalo_mlp_out': ('/mlp/wo/',),                       # A = gated-GELU output g (K = mlp_dim)
    'f16_noalo_attn_out': ('/out/kernel',),                   # A = attention output (self and cross)
    'f16_noalo_mlp_in': ('/mlp/wi_',),                        # A = normalised residual y
    'f16_noalo_qkv': ('/query/', '/key/', '/value/'),         # A = normalised residual y / encodings
    'f16_noalo_outs': ('/mlp/wo/', '/out/kernel'),
}
def _f16(x):
  import torch
  return x.clamp(-65504.0, 65504.0).to(torch.float16).to(x.dtype)


def _split_f16(x):
  hi = _f16(x)
  return (hi, _f16(x - hi))


class StudyModel(fast.FastModel):
  variant = 'x3'

  def _split(self, a, which='a'):
    if self.variant in F16_VARIANTS:
      return _split_f16(a)
    parts = super()._split(a)
    if self.variant in LO8_VARIANTS and len(parts) == 2:
      who, kind = LO8_VARIANTS[self.variant]
      if which in who:
        lo = _fp8_round(self.xp, a - parts[0], kind)   # quantise the exact residual, not its bf16 rounding
        return (parts[0], lo)
    return parts

  def _w(self, name):
    if self.variant in F16_VARIANTS:
      if name not in self._wcache:
        import torch
        w = self.p[name]
        sc = 1.0
        if F16_VARIANTS[self.variant]['scale_w']:
          sc = float(2.0 ** torch.floor(torch.log2(16384.0 / w.abs().max())))   # max |w| lands in [8192, 16384)
        self._wcache[name] = (_split_f16(w * sc), sc)
      return self._wcache[name]
    if name not in self._wcache:
      # weights [K, N]: the device stores W^T rows of K, so the scale runs along K = axis 0 here
      w = self.p[name]
      if self.variant in LO8_VARIANTS and 'w' in LO8_VARIANTS[self.variant][0]:
        hi = self.xp.round_bf16(w)
        lo = _fp8_round(self.xp, (w - hi).T, LO8_VARIANTS[self.variant][1]).T
        self._wcache[name] = ((hi, lo), 1.0)
      else:
        self._wcache[name] = (fast.FastModel._split(self, w), 1.0)
    return self._wcache[name]

  def mm(self, a, name):
    self._drop_alo = self.variant in F16_ALO_DROP and any(sub in name for sub in F16_ALO_DROP[self.variant])
    return super().mm(a, name)

  def _mm_parts(self, a_parts, w_parts):
    if self.variant in F16_VARIANTS:
      if getattr(self, '_drop_alo', False):
        return self.xp.matmul(a_parts[0], w_parts[0]) + self.xp.matmul(a_parts[0], w_parts[1])
      y = 0
      for a, b in F16_VARIANTS[self.variant]['mm']:
        y = y + self.xp.matmul(a_parts[a], w_parts[b])
      return y            # FastModel.mm undoes the weight scale
    if self.variant not in MM_VARIANTS:
      return super()._mm_parts(a_parts, w_parts)
    y = 0
    for a, b in MM_VARIANTS[self.variant]:
      y = y + self.xp.matmul(a_parts[a], w_parts[b])
    return y

  def decoder_pass(self, z, i, cond):
    self._in_decoder = True
    try:
      return super().decoder_pass(z, i, cond)
    finally:
      self._in_decoder = False

  def _attend(self, q, k, v):
    xp = self.xp
    qk, pv = VARIANTS.get(self.variant, VARIANTS['x3'])
    if self.variant in F16_ATT:
      qk, pv, where = F16_ATT[self.variant]
      # decoder self-attention has as many keys as queries (and encoder self-attention too); cross-attention
      # is the only call with a different key count on these configs
      is_self = q.shape[0] == k.shape[0]
      if (where == 'self' and not is_self) or (where == 'cross' and is_self) or (
          where == 'dec' and not getattr(self, '_in_decoder', False)):
        qk, pv = _X3, _X3
    def split(a):
      if self.variant in F16_VARIANTS:
        return _split_f16(a)
      hi = xp.round_bf16(a)
      return (hi, xp.round_bf16(a - hi))
    qs, ks, vs = split(q), split(k), split(v)
    s = 0
    for a, b in qk:
      s = s + xp.einsum('qhd,khd->hqk', qs[a], ks[b])
    m = xp.max(s, axis=-1, keepdims=True)
    pr = xp.exp(s - m)
    l = xp.sum(pr, axis=-1, keepdims=True)
    ps = split(pr)
    o = 0
    for a, b in pv:
      o = o + xp.einsum('hqk,khd->hqd', ps[a], vs[b])
    o = xp.einsum('hqd->qhd', o / l)
    return xp.reshape(o, (q.shape[0], q.shape[1] * q.shape[2]))


def base_song(names):
  """The two chained base_with_context golden segments (tests/golden/make_golden.py: song)."""
  g = np.load(os.path.join(GOLD, 'base_with_context_n1000.npz'))
  spec = msd_amd.config.preset('base_with_context', num_steps=1000)
  params = msd_amd.synthetic.init_params(spec, 0)
  cfg, dc = helpers.oracle_configs(spec)
  t, n, c = 256, 128, 256
  for name in names:
    xp = backend.TorchBackend('float32', threads=backend.effective_cpus())
    m = StudyModel(xp, cfg, dc, params, True, precision='bf16x3')
    m.variant = name
    pred = np.zeros((1, c, n), np.float32)
    for k in range(int(g['n_segments'])):
      t0 = time.perf_counter()
      batch = {'encoder_input_tokens': msd_amd.synthetic.segment_tokens(spec, k), 'encoder_continuous_inputs': pred,
               'encoder_continuous_mask': (np.zeros if k == 0 else np.ones)((1, c), np.int32)}
      init_z, noise = philox.segment_noise((1, t, n), 1000, seed=int(g['noise_seed']), segment=k)
      pred = xp.to_numpy(m.predict(batch, init_z, noise)[0]).astype(np.float32)
      print('base %-10s segment %d rms vs float64 golden %.3e   (%.0f s)'
            % (name, k, helpers.rms(pred, g['mel'][:, k * t:(k + 1) * t]), time.perf_counter() - t0), flush=True)


def main(names):
  if names and names[0] == 'base':
    return base_song(names[1:] or ['x3'])
  g = np.load(os.path.join(GOLD, 'small_n1000.npz'))
  spec = msd_amd.config.preset('small', num_steps=1000)
  params = msd_amd.synthetic.init_params(spec, 0)
  cfg, dc = helpers.oracle_configs(spec)
  t, n = spec.task_feature_lengths['targets'], 128
  batch = {'encoder_input_tokens': msd_amd.synthetic.segment_tokens(spec, 0)}
  init_z, noise = philox.segment_noise((1, t, n), 1000, seed=int(g['noise_seed']), segment=0)
  for name in names:
    xp = backend.TorchBackend('float32', threads=backend.effective_cpus())
    m = StudyModel(xp, cfg, dc, params, spec.has_context, precision='bf16x3')
    m.variant = name
    t0 = time.perf_counter()
    out = xp.to_numpy(m.predict(batch, init_z, noise)[0])
    print('%-10s rms vs float64 golden %.3e   (%.0f s)' % (name, helpers.rms(out[:, :t], g['mel'][:, :t]), time.perf_counter() - t0), flush=True)


if __name__ == '__main__':
  main(sys.argv[1:] or list(VARIANTS) + list(MM_VARIANTS) + list(LO8_VARIANTS) + list(F16_VARIANTS))
