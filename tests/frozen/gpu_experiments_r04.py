"""GPU parity tests of the EXPERIMENTS build (tools/ubench/exp/libmsd_amd_exp.so, -DMSD_EXPERIMENTS=1): kernels and launch
structures that were measured and rejected -- or are this round's experiment -- stay parity-tested as long as their
numbers are quoted (docs/history.md).  The product libraries contain none of them and read no
environment variable; these tests load the experiments library through MSD_AMD_LIB and skip when it has not been
built (python music-spectrogram-diffusion_amd/build_native.py --experiments)."""
import os

import numpy as np
import pytest

import msd_amd
from msd_amd import native
from tests import helpers
from tests.test_gpu_model import _oracle   # not collected by `pytest tests/` (file name); run: python -m pytest tests/frozen/gpu_experiments_r04.py -m gpu

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
EXP_LIB = os.path.join(ROOT, 'tools', 'ubench', 'exp', 'libmsd_amd_exp.so')


@pytest.fixture
def exp_lib(monkeypatch):
  """The half-plane library of this test is the experiments build; the product library comes back afterwards."""
  if not os.path.exists(EXP_LIB):
    pytest.skip('experiments library not built')
  monkeypatch.setenv('MSD_AMD_LIB', EXP_LIB)
  monkeypatch.setattr(native, '_libs', {})
  return monkeypatch


@pytest.mark.parametrize('switch', ['MSD_BIG_PAIR', 'MSD_BIG_WIDE', 'MSD_BIG_LS'])
def test_batched_tile_variants_match_oracle(switch, exp_lib):
  """16 songs per handle: M = 2*16*64 = 2048 rows -> the 128-row GEMM tiles of the batched
  path (msd_api.hip big_m_threshold).  emb 192 / 3 heads / mlp 256 make every N a multiple of the
  96/128-column tiles so all big instantiations run; checked per song against the oracle.
  `switch`: the batched path's alternative tile kernels, off by default (built, parity-green, not
  faster) -- K = 32 tiles with two blocks per CU (gemm_h16_pair.h, all five 128-row launches), the 256 x 128
  eight-wave tile and the 256 x 128 tile with loader waves (gemm_h16_wide.h / gemm_h16_ls.h, gated-MLP input)."""
  import dataclasses
  exp_lib.setenv(switch, '1')   # read by msd_create of the experiments build
  base = msd_amd.config.preset('tiny_context', num_steps=4)
  spec = dataclasses.replace(base, t5=dataclasses.replace(base.t5, emb_dim=192, num_heads=3))
  params = msd_amd.synthetic.init_params(spec, 5, norm_scale_jitter=0.1)
  B = 16
  model = msd_amd.InferenceModel(params, spec, batch_size=B, **helpers.ALL_PLANES)
  batch = helpers.make_batch(spec, batch=B, ctx_mask='ragged')
  init_z, noise = helpers.make_noise(spec, batch=B)
  got, _ = model.predict(batch, init_z=init_z, noise=noise)
  ref64, _ = _oracle(spec, params, batch, init_z, noise, 'float64')
  ref32, _ = _oracle(spec, params, batch, init_z, noise, 'float32')
  helpers.assert_fp32_class(got, ref64, ref32, what='batched B=16')
  # and one decoder pass, elementwise (no chaotic amplification)
  import torch
  from oracle import backend, fast
  cfg, dc = helpers.oracle_configs(spec)
  fm = fast.FastModel(backend.NumpyBackend('float64'), cfg, dc, params, True)
  fm.encode(batch['encoder_input_tokens'], batch['encoder_continuous_inputs'], batch['encoder_continuous_mask'])
  nm = model._get_native()
  z = np.random.default_rng(1).standard_normal((B, 64, 128)).astype(np.float32)
  zd = torch.as_tensor(z).cuda()
  for step, cond in [(3, True), (1, False)]:
    eps = torch.zeros_like(zd)
    nm.decoder_pass(B, step, zd, cond, eps)
    torch.cuda.synchronize()
    want = fm.decoder_pass(z.astype(np.float64), step, cond)
    err = np.abs(eps.cpu().numpy() - want).max() / np.abs(want).max()
    assert err < 2e-4, (step, cond, err)



@pytest.mark.parametrize('mode', ['1', '2'])
def test_xcd_resident_chain_kernel_matches_separate_launches(exp_lib, mode):
  """MSD_CHAIN=1 / 2: MLP-in -> MLP-out -> next layer's QKV as ONE launch whose phases are separated by XCD-local
  barriers (tools/ubench/exp/chain.h; 2 = with the next phase's weight tiles pre-staged before the barrier, round 4).
  Same tiles, same arithmetic: the eps of a decoder pass and a whole sampled segment must agree with the
  separate-launch path of the SAME library."""
  import torch
  spec = msd_amd.config.preset('tiny_context', num_steps=6)
  params = msd_amd.synthetic.init_params(spec, 3, norm_scale_jitter=0.1)
  batch = helpers.make_batch(spec, batch=2, ctx_mask='ragged')
  init_z, noise = helpers.make_noise(spec, batch=2)
  outs, eps = {}, {}
  for chain in ('0', mode):
    exp_lib.setenv('MSD_CHAIN', chain)
    model = msd_amd.InferenceModel(params, spec, batch_size=2, **helpers.ALL_PLANES)
    outs[chain], _ = model.predict(batch, init_z=init_z, noise=noise)
    nm = model._get_native()
    z = torch.as_tensor(init_z).cuda()
    e = torch.zeros_like(z)
    nm.decoder_pass(2, 3, z, True, e)
    torch.cuda.synchronize()
    eps[chain] = e.cpu().numpy()
  rel = np.abs(eps[mode] - eps['0']).max() / np.abs(eps['0']).max()
  print('chain vs separate launches: decoder pass max rel diff %.2e' % rel)
  assert rel < 1e-5
  ref64, _ = _oracle(spec, params, batch, init_z, noise, 'float64')
  ref32, _ = _oracle(spec, params, batch, init_z, noise, 'float32')
  helpers.assert_fp32_class(outs[mode], ref64, ref32, 'chain')


@pytest.mark.parametrize('preset,mask', [('tiny_context', 'ragged'), ('tiny_context', 'zeros'), ('tiny', 'ones')])
def test_hoisted_cross_query_projection_matches_the_plain_order(exp_lib, preset, mask):
  """MSD_HOIST_Q=1 (experiments build): the cross-attention query projection runs in the launch of the self-attention
  output projection, on [x0 (.) gamma | attention output] . [Wq ; Wo diag(gamma) Wq], and the RMSNorm's 1/rms is
  applied to the logits inside the attention kernel (csrc/msd_api.hip decoder_layers).  Exact algebra, different
  rounding order: single decoder passes must agree with the un-hoisted order far inside the float32 class, both
  must sit on the float64 oracle, and a sampled segment stays in the float32 class."""
  import torch
  from oracle import backend, fast
  spec = msd_amd.config.preset(preset, num_steps=6)
  params = msd_amd.synthetic.init_params(spec, 11, norm_scale_jitter=0.3)
  batch = helpers.make_batch(spec, batch=2, ctx_mask=mask) if spec.has_context else helpers.make_batch(spec, batch=2)
  init_z, noise = helpers.make_noise(spec, batch=2)
  cfg, dc = helpers.oracle_configs(spec)
  xp = backend.TorchBackend('float64')
  fm = fast.FastModel(xp, cfg, dc, params, spec.has_context)
  if spec.has_context:
    fm.encode(batch['encoder_input_tokens'], batch['encoder_continuous_inputs'], batch['encoder_continuous_mask'])
  else:
    fm.encode(batch['encoder_input_tokens'])
  outs, eps = {}, {}
  for hoist in ('0', '1'):
    exp_lib.setenv('MSD_HOIST_Q', hoist)
    model = msd_amd.InferenceModel(params, spec, batch_size=2, **helpers.ALL_PLANES)
    outs[hoist], _ = model.predict(batch, init_z=init_z, noise=noise)
    nm = model._get_native()
    z = torch.as_tensor(init_z).cuda()
    e = torch.zeros_like(z)
    for step in (5, 0):
      nm.decoder_pass(2, step, z, True, e)
      torch.cuda.synchronize()
      eps[hoist, step] = e.cpu().numpy().astype(np.float64)
  for step in (5, 0):
    ref = xp.to_numpy(fm.decoder_pass(xp.asarray(init_z), step, True)).astype(np.float64)
    rel = np.abs(eps['1', step] - eps['0', step]).max() / np.abs(eps['0', step]).max()
    e1 = np.abs(eps['1', step] - ref).max() / np.abs(ref).max()
    e0 = np.abs(eps['0', step] - ref).max() / np.abs(ref).max()
    print('%s/%s step %d: hoisted vs plain %.2e; vs float64 oracle: hoisted %.2e, plain %.2e' % (preset, mask, step, rel, e1, e0))
    assert rel < 5e-5 and e1 < 2e-4 and e0 < 2e-4
  ref64, _ = _oracle(spec, params, batch, init_z, noise, 'float64')
  ref32, _ = _oracle(spec, params, batch, init_z, noise, 'float32')
  helpers.assert_fp32_class(outs['1'], ref64, ref32, 'hoisted query projection')


@pytest.mark.parametrize('film', [True, False])
@pytest.mark.parametrize('m,k,d,n', [(512, 2048, 768, 2304), (256, 2048, 768, 768), (512, 1024, 512, 1536)])
def test_split_k_producer_of_the_folded_norm(exp_lib, film, m, k, d, n):
  """msd_op_residual_norm_gemm(folded=2): the folded RMSNorm + FiLM path with its producer on the 4-way split-K launch
  (three launches over the same arrival counters) against the float64 oracle and the plain producer."""
  import torch
  from oracle import backend, ops
  xp = backend.NumpyBackend('float64')
  rng = np.random.default_rng(m + k + d + n)
  x_in = (3.0 * rng.standard_normal((m, d))).astype(np.float32)
  x_in[:, ::7] *= 20.0
  a = rng.standard_normal((m, k)).astype(np.float32)
  w1 = (rng.standard_normal((k, d)) / np.sqrt(k)).astype(np.float32)
  gamma = (1.0 + 0.3 * rng.standard_normal(d)).astype(np.float32)
  sc = (0.5 * rng.standard_normal(d)).astype(np.float32) if film else None
  bi = (0.5 * rng.standard_normal(d)).astype(np.float32) if film else None
  w2 = (rng.standard_normal((d, n)) / np.sqrt(d)).astype(np.float32)
  x_ref = x_in.astype(np.float64) + a.astype(np.float64) @ w1.astype(np.float64)
  h = ops.rms_layer_norm(xp, x_ref, gamma.astype(np.float64))
  if film:
    h = h * (sc.astype(np.float64) + 1.0) + bi.astype(np.float64)
  h_ref = h @ w2.astype(np.float64)
  dev = lambda v: torch.as_tensor(np.ascontiguousarray(v)).cuda()
  res = {}
  for folded in (1, 2):
    x_out = torch.empty((m, d), dtype=torch.float32, device='cuda')
    h_out = torch.empty((m, n), dtype=torch.float32, device='cuda')
    native.op_residual_norm_gemm(folded, dev(x_in), dev(a), dev(w1), dev(gamma), None if sc is None else dev(sc),
                                 None if bi is None else dev(bi), dev(w2), x_out, h_out)
    res[folded] = (x_out.cpu().numpy(), h_out.cpu().numpy())
    ex = np.abs(res[folded][0] - x_ref).max() / np.abs(x_ref).max()
    eh = np.abs(res[folded][1] - h_ref).max() / np.abs(h_ref).max()
    assert ex < 2e-5 and eh < 4e-5, (folded, ex, eh)
  np.testing.assert_allclose(res[2][0], res[1][0], rtol=0, atol=1e-5 * np.abs(x_ref).max())
