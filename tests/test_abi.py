"""The C-ABI library builds, loads and exports exactly what include/msd_amd.h
declares (no compute calls: this runs without a GPU)."""
import ctypes
import os
import re

import pytest

import msd_amd
from msd_amd import native

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _header_functions():
  text = open(os.path.join(ROOT, 'include', 'msd_amd.h')).read()
  text = re.sub(r'/\*.*?\*/', '', text, flags=re.S)
  return sorted(set(re.findall(r'\b(msd_[a-z0-9_]+)\s*\(', text)))


@pytest.fixture(scope='module')
def lib():
  import __graft_entry__
  __graft_entry__.build()
  return native.load()


def test_header_and_binding_agree():
  assert _header_functions() == sorted(native.EXPORTED_SYMBOLS)


def test_library_exports_every_declared_symbol(lib):
  for name in _header_functions():
    assert hasattr(lib, name), name
  assert b'gfx950' in lib.msd_version() and b'half planes' in lib.msd_version()


def test_bfloat16_plane_build_exports_the_same_abi(lib):
  other = native.load('bf16')
  for name in _header_functions():
    assert hasattr(other, name), name
  assert b'bfloat16 planes' in other.msd_version()
  assert native.plane_format('f16x3') == 'f16' and native.plane_format('bf16x3') == 'bf16'
  with pytest.raises(ValueError):
    native.plane_format('fp8')


def test_config_struct_layout_matches_header(lib):
  text = open(os.path.join(ROOT, 'include', 'msd_amd.h')).read()
  body = text[text.index('typedef struct msd_config {'):text.index('} msd_config;')]
  fields = re.findall(r'^\s*(int32_t|float)\s+(\w+);', body, flags=re.M)
  assert [(n, {'int32_t': ctypes.c_int32, 'float': ctypes.c_float}[t]) for t, n in fields] == \
      list(native.MsdConfig._fields_)


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
  monkeypatch.setattr(native, '_libs', {})
  monkeypatch.setattr(native, 'LIB_PATHS', {'f16': str(tmp_path / 'nope.so'), 'bf16': str(tmp_path / 'nope2.so')})
  with pytest.raises(native.NativeLibraryError):
    native.load()
  with pytest.raises(native.NativeLibraryError):
    native.load('bf16')


def test_product_never_imports_the_oracle():
  pkg = os.path.join(ROOT, 'music-spectrogram-diffusion_amd')
  for dirpath, _, files in os.walk(pkg):
    for f in files:
      if f.endswith(('.py', '.h', '.hip', '.cpp')):
        src = open(os.path.join(dirpath, f)).read()
        assert not re.search(r'^\s*(from|import)\s+oracle\b', src, flags=re.M), f


def test_integration_stub_declares_the_same_struct():
  """INTEGRATION.md shows the ctypes stub a reference maintainer would add: its MsdConfig must list the fields of
  the header, in order (the stub is documentation, but a stale one would bind garbage)."""
  text = open(os.path.join(ROOT, 'INTEGRATION.md')).read()
  body = text[text.index('class MsdConfig(ctypes.Structure):'):text.index('SCHEDULE = {')]
  names = re.findall(r"'([a-z_0-9]+)'", body)
  assert names == [n for n, _ in native.MsdConfig._fields_]
