"""T5X checkpoint directories (msgpack index + zarr arrays).  No reference checkpoint
is reachable here, so the reader is pinned by the writer (same published layout), by hand-built zarr
arrays (chunked, ragged edge chunks, zlib / gzip / none, 0-d, missing chunk -> fill value) and by the
parameter-tree check against config.param_shapes.  CPU only."""
import gzip
import json
import os

import numpy as np
import pytest

import msd_amd
from msd_amd import checkpoints


def test_zarr_chunked_ragged_and_compressors(tmp_path):
  rng = np.random.default_rng(0)
  a = rng.standard_normal((10, 7)).astype(np.float32)
  for comp in ('gzip', 'zlib', None):
    d = str(tmp_path / ('arr_%s' % comp))
    checkpoints.write_zarr_array(d, a, compressor=comp, chunks=(4, 3))       # 3 x 3 chunks, ragged edges
    assert sorted(n for n in os.listdir(d) if n != '.zarray')[:3] == ['0.0', '0.1', '0.2']
    np.testing.assert_array_equal(checkpoints.read_zarr_array(d), a)
  # 0-d array and a missing chunk
  d0 = str(tmp_path / 'scalar')
  checkpoints.write_zarr_array(d0, np.float32(3.5))
  assert checkpoints.read_zarr_array(d0) == np.float32(3.5)
  d = str(tmp_path / 'arr_gzip')
  os.remove(os.path.join(d, '2.2'))
  meta = json.load(open(os.path.join(d, '.zarray')))
  meta['fill_value'] = 9.0
  json.dump(meta, open(os.path.join(d, '.zarray'), 'w'))
  got = checkpoints.read_zarr_array(d)
  assert (got[8:, 6:] == 9.0).all() and np.array_equal(got[:8], a[:8])


def test_zarr_hand_built_bytes(tmp_path):
  """A chunk file is the gzip of the C-order bytes of a FULL chunk (zarr v2 spec)."""
  d = tmp_path / 'x'
  d.mkdir()
  (d / '.zarray').write_text(json.dumps({'chunks': [2, 2], 'compressor': {'id': 'gzip', 'level': 1}, 'dtype': '<f4',
                                         'fill_value': None, 'filters': None, 'order': 'C', 'shape': [2, 3],
                                         'zarr_format': 2}))
  (d / '0.0').write_bytes(gzip.compress(np.array([[1, 2], [4, 5]], '<f4').tobytes()))
  (d / '0.1').write_bytes(gzip.compress(np.array([[3, 0], [6, 0]], '<f4').tobytes()))   # padded edge chunk
  np.testing.assert_array_equal(checkpoints.read_zarr_array(str(d)), [[1, 2, 3], [4, 5, 6]])
  with pytest.raises(checkpoints.CheckpointError):
    (d / '.zarray').write_text(json.dumps({'chunks': [2, 2], 'compressor': {'id': 'blosc'}, 'dtype': '<f4',
                                           'fill_value': None, 'filters': None, 'order': 'C', 'shape': [2, 3],
                                           'zarr_format': 2}))
    checkpoints.read_zarr_array(str(d))


@pytest.mark.parametrize('inline_below', [0, 300])
def test_t5x_round_trip_matches_param_tree(tmp_path, inline_below):
  spec = msd_amd.config.preset('tiny_context')
  params = msd_amd.synthetic.init_params(spec, 1, norm_scale_jitter=0.1)
  ckpt = checkpoints.save_t5x_checkpoint(params, str(tmp_path / 'model'), step=1234, inline_below=inline_below)
  assert os.path.basename(ckpt) == 'checkpoint_1234'
  assert os.path.isdir(os.path.join(ckpt, 'target.decoder.layers_0.FiLMLayer_0.DenseGeneral_0.kernel')) or inline_below
  for path in (ckpt, str(tmp_path / 'model')):            # specific checkpoint, or the model dir (latest step)
    got = checkpoints.load_t5x_checkpoint(path)
    assert int(got.pop('__step__')) == 1234
    shapes = msd_amd.config.param_shapes(spec)
    assert set(got) == set(shapes) == set(params)
    for k, v in got.items():
      assert v.dtype == np.float32 and v.shape == tuple(shapes[k])
      np.testing.assert_array_equal(v, params[k])
  # the index is optional: plain directory scan of target.* (only when nothing was inlined)
  if not inline_below:
    os.remove(os.path.join(ckpt, 'checkpoint'))
    got = checkpoints.load_t5x_checkpoint(ckpt)
    got.pop('__step__')
    assert set(got) == set(params)


def test_latest_step_and_errors(tmp_path):
  p = {'a/b': np.ones((2, 2), np.float32)}
  checkpoints.save_t5x_checkpoint(p, str(tmp_path / 'm'), step=10)
  p2 = {'a/b': np.full((2, 2), 2, np.float32)}
  checkpoints.save_t5x_checkpoint(p2, str(tmp_path / 'm'), step=200)
  got = checkpoints.load_t5x_checkpoint(str(tmp_path / 'm'))
  assert int(got['__step__']) == 200 and (got['a/b'] == 2).all()
  (tmp_path / 'empty').mkdir()
  with pytest.raises(checkpoints.CheckpointError):
    checkpoints.load_t5x_checkpoint(str(tmp_path / 'empty'))


def test_step_comes_from_the_saved_train_state_not_the_directory_name(tmp_path):
  """The reference reads train_state.step (inference.py:178-181): a renamed / copied checkpoint
  directory keeps reporting the step it was saved at."""
  import shutil
  p = {'a/b': np.ones((2, 2), np.float32)}
  ckpt = checkpoints.save_t5x_checkpoint(p, str(tmp_path / 'm'), step=1234)
  shutil.move(ckpt, str(tmp_path / 'renamed_copy'))
  got = checkpoints.load_t5x_checkpoint(str(tmp_path / 'renamed_copy'))
  assert int(got['__step__']) == 1234


def test_inference_loader_accepts_t5x_dir(tmp_path):
  """InferenceModel's restore path (reference inference.py:159-176 takes the checkpoint directory)."""
  from msd_amd import inference
  spec = msd_amd.config.preset('tiny')
  params = msd_amd.synthetic.init_params(spec, 2)
  ckpt = checkpoints.save_t5x_checkpoint(params, str(tmp_path / 'm'), step=77)
  got, step = inference._load_checkpoint(ckpt, spec)
  assert step == 77 and set(got) == set(params)
  assert all(np.array_equal(got[k], params[k]) for k in params)
  with pytest.raises(ValueError):
    inference._load_checkpoint(str(tmp_path / 'nope.bin'), spec)


@pytest.mark.gpu
def test_device_model_from_t5x_checkpoint(tmp_path):
  from tests import helpers
  spec = msd_amd.config.preset('tiny_context', num_steps=4)
  params = msd_amd.synthetic.init_params(spec, 3, norm_scale_jitter=0.1)
  ckpt = checkpoints.save_t5x_checkpoint(params, str(tmp_path / 'm'), step=5)
  a = msd_amd.InferenceModel(ckpt, spec)
  b = msd_amd.InferenceModel(params, spec)
  batch = helpers.make_batch(spec)
  init_z, noise = helpers.make_noise(spec)
  ya, _ = a.predict(batch, init_z=init_z, noise=noise)
  yb, _ = b.predict(batch, init_z=init_z, noise=noise)
  assert a.step == 5 and np.array_equal(ya, yb)


def test_zarr_round_trip_property(tmp_path):
  """Random ranks / shapes / chunkings / compressors (incl. chunks larger than the array)."""
  from hypothesis import given, settings, strategies as st
  counter = [0]

  @settings(max_examples=40, deadline=None)
  @given(st.lists(st.integers(1, 7), min_size=0, max_size=3), st.data())
  def check(shape, data):
    chunks = [data.draw(st.integers(1, s + 2)) for s in shape]
    comp = data.draw(st.sampled_from(['gzip', 'zlib', None]))
    arr = np.arange(int(np.prod(shape)) if shape else 1, dtype=np.float32).reshape(shape) * 0.5 - 3
    counter[0] += 1
    d = str(tmp_path / ('z%d' % counter[0]))
    checkpoints.write_zarr_array(d, arr, compressor=comp, chunks=chunks if shape else None)
    np.testing.assert_array_equal(checkpoints.read_zarr_array(d), arr)
  check()


def test_t5x_directory_built_with_msgpack_and_raw_zarr_not_with_our_writer(tmp_path):
  """The reader against a checkpoint directory this package's WRITER never touched: the index is
  packed with the `msgpack` library exactly as flax.serialization / t5x.checkpoints do it --
    * {'version': 3, 'optimizer': {'state': {'step': ..., 'param_states': {...}}, 'target': {...}}}
    * a small array inline as flax's ExtType 1 = packb((shape, dtype.name, bytes)), the step as an int32 0-d array
    * a large array as the TensorStore spec t5x stores in its place ({'driver': 'zarr', 'kvstore': {'driver': 'file',
      'path': 'target.<dotted name>'}, 'metadata': {...}, 'dtype': ...}), with the optimizer's slots of the same
      parameters stored the same way under 'state.param_states.*' (they must be skipped)
  -- and the arrays are raw zarr v2 directories laid out by hand the way t5x writes them: several chunks ([chunk, full]
  row blocks), gzip-compressed C-order bytes of FULL chunks (ragged last block padded), '.zarray' JSON per the zarr
  v2 spec.  Names, shapes, values and the step must come out."""
  import msgpack
  rng = np.random.default_rng(5)
  ckpt = tmp_path / 'model' / 'checkpoint_5000'
  ckpt.mkdir(parents=True)

  def zarr_dir(rel, arr, rows_per_chunk):
    d = ckpt / rel
    d.mkdir()
    chunks = [rows_per_chunk] + list(arr.shape[1:])
    (d / '.zarray').write_text(json.dumps({
        'chunks': chunks, 'compressor': {'id': 'gzip', 'level': 1}, 'dtype': arr.dtype.str, 'fill_value': None,
        'filters': None, 'order': 'C', 'shape': list(arr.shape), 'zarr_format': 2}))
    for i in range(-(-arr.shape[0] // rows_per_chunk)):
      block = np.zeros(chunks, arr.dtype)
      part = arr[i * rows_per_chunk:(i + 1) * rows_per_chunk]
      block[:len(part)] = part
      key = '.'.join([str(i)] + ['0'] * (arr.ndim - 1))
      (d / key).write_bytes(gzip.compress(block.tobytes(), 1))
    return {'driver': 'zarr', 'dtype': arr.dtype.name, 'kvstore': {'driver': 'file', 'path': rel},
            'metadata': {'chunks': chunks, 'compressor': {'id': 'gzip'}, 'shape': list(arr.shape)}}

  def inline(arr):
    arr = np.asarray(arr)
    return msgpack.ExtType(1, msgpack.packb((list(arr.shape), arr.dtype.name, arr.tobytes()), use_bin_type=True))

  kernel = rng.standard_normal((70, 48)).astype(np.float32)           # 3 row chunks of 32, ragged last
  embed = rng.standard_normal((20, 16)).astype(np.float32)
  scale = rng.standard_normal((48,)).astype(np.float32)
  state = {
      'version': 3,
      'optimizer': {
          'state': {'step': inline(np.asarray(5000, np.int32)),
                    'param_states': {'decoder': {'layers_0': {'mlp': {'wo': {'kernel': {
                        'v_row': zarr_dir('state.param_states.decoder.layers_0.mlp.wo.kernel.v_row', np.ones((70,), np.float32), 64)}}}}}}},
          'target': {
              'decoder': {'layers_0': {'mlp': {'wo': {'kernel': zarr_dir('target.decoder.layers_0.mlp.wo.kernel', kernel, 32)}},
                                       'pre_mlp_layer_norm': {'scale': inline(scale)}}},
              'token_encoder': {'token_embedder': {'embedding': zarr_dir('target.token_encoder.token_embedder.embedding', embed, 8)}},
          },
      },
  }
  (ckpt / 'checkpoint').write_bytes(msgpack.packb(state, use_bin_type=True))
  for path in (str(ckpt), str(tmp_path / 'model')):
    got = checkpoints.load_t5x_checkpoint(path)
    assert int(got.pop('__step__')) == 5000
    assert set(got) == {'decoder/layers_0/mlp/wo/kernel', 'decoder/layers_0/pre_mlp_layer_norm/scale',
                        'token_encoder/token_embedder/embedding'}
    np.testing.assert_array_equal(got['decoder/layers_0/mlp/wo/kernel'], kernel)
    np.testing.assert_array_equal(got['decoder/layers_0/pre_mlp_layer_norm/scale'], scale)
    np.testing.assert_array_equal(got['token_encoder/token_embedder/embedding'], embed)
    assert all(v.dtype == np.float32 for v in got.values())
  # a renamed directory still reports the step saved INSIDE the index (the reference reads train_state.step)
  os.rename(str(ckpt), str(tmp_path / 'model' / 'copy_of_it'))
  assert int(checkpoints.load_t5x_checkpoint(str(tmp_path / 'model' / 'copy_of_it'))['__step__']) == 5000
