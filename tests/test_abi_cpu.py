"""The C-ABI library must load without a GPU and export every symbol include/geeco_hip.h declares
(no compute calls here)."""
import ctypes
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
  src = open(os.path.join(ROOT, 'include', 'geeco_hip.h')).read()
  src = re.sub(r'/\*.*?\*/', '', src, flags=re.S)
  return sorted(set(re.findall(r'\b(geeco_[a-z0-9_]+)\s*\(', src)))


def test_library_exports_every_declared_symbol():
  from geeco_amd import _native
  names = _declared()
  assert len(names) >= 25
  lib = ctypes.CDLL(_native.LIB_PATH)
  missing = [n for n in names if not hasattr(lib, n)]
  assert not missing, missing
  # the Python binding types exactly the declared set
  assert sorted(_native.SIGNATURES.keys()) == names


def test_binding_loads_and_reports_version():
  from geeco_amd import _native
  lib = _native.load()
  hdr = open(os.path.join(ROOT, 'include', 'geeco_hip.h')).read()
  declared = int(re.search(r'#define GEECO_ABI_VERSION (\d+)', hdr).group(1))
  assert lib.geeco_abi_version() == declared == _native.ABI_VERSION     # header, library and binding agree
  assert lib.geeco_dynimg_ws_bytes(2, 1024) > 0
  assert lib.geeco_conv3x3_wgrad_ws_bytes(3, 2, 64, 64, 32, 48, 2) > 0
  assert lib.geeco_conv3x3_fwd_ws_bytes(3, 32, 256, 256, 4, 32, 1) == 0       # big layers never split K
  assert lib.geeco_conv3x3_fwd_ws_bytes(3, 32, 4, 4, 256, 256, 2) > 0         # conv8 does


def test_host_side_argument_checks_need_no_gpu():
  """Bad arguments are rejected before any launch, with a message (error behaviour of the boundary)."""
  from geeco_amd import _native
  lib = _native.load()
  rc = lib.geeco_conv3x3_fwd(None, None, None, None, 1, 0, 0, 0, 0, 1, 8, 8, 4, 16, 1, 1, None, None)
  assert rc == -1 and b'null pointer' in lib.geeco_last_error()
  one = ctypes.c_void_p(16)
  rc = lib.geeco_conv3x3_fwd(one, one, one, one, 1, 0, 0, 0, 0, 1, 8, 8, 3, 16, 1, 1, None, None)
  assert rc == -1 and b'multiple of 4' in lib.geeco_last_error()
  rc = lib.geeco_dynimg_fwd(one, None, 0, 0, one, 1, 65, 16, 3, 4, one, one, None)
  assert rc == -1 and b'K=65' in lib.geeco_last_error()
  # the batched LSTM-step backward: null operands, a concat description without its arrays, a state narrower than its cells
  rc = lib.geeco_lstm_step_bwd(None, 8, one, 8, one, 8, one, 8, one, one, 8, 2, 8, 8, None, None, None, 0, 0, 0, 0, one, None, None)
  assert rc == -1 and b'null pointer' in lib.geeco_last_error()
  rc = lib.geeco_lstm_step_bwd(one, 8, one, 8, one, 8, one, 8, one, one, 8, 2, 8, 8, None, None, None, 2, 0, 0, 4, one, None, None)
  assert rc == -1 and b'concat description' in lib.geeco_last_error()
  arr = (ctypes.c_void_p * 1)(16)
  chs = (ctypes.c_int * 1)(256)
  rc = lib.geeco_lstm_step_bwd(one, 8, one, 8, one, 8, one, 8, one, one, 8, 2, 8, 8, arr, arr, chs, 1, 1, 7, 4, one, None, None)
  assert rc == -1 and b'exceed the state width' in lib.geeco_last_error()
  assert lib.geeco_lstm_step_bwd_ws_bytes(32, 3100, 512) == lib.geeco_gemm_ws_bytes(32, 3100, 512) > 0


def test_alpha_matches_oracle():
  from geeco_amd import _native
  from oracle import geeco_oracle as O
  lib = _native.load()
  for K in (2, 4, 16, 32):
    buf = (ctypes.c_float * K)()
    lib.geeco_dynimg_alpha(K, ctypes.cast(buf, ctypes.c_void_p))
    assert [float(v) for v in buf] == [float(v) for v in O.dynimg_alpha(K)]    # bit-exact float32


def test_stale_library_is_refused_by_version(tmp_path, monkeypatch):
  """A library of another ABI version (a stale build, or another build picked through GEECO_LIB) must be refused at load
  time with a message that says so -- not load and fail later at the first changed entry point."""
  import subprocess
  import pytest
  from geeco_amd import _native
  src = tmp_path / 'stale.c'
  src.write_text('int geeco_abi_version(void) { return 1; }\n')
  so = tmp_path / 'libstale.so'
  subprocess.check_call(['gcc', '-shared', '-fPIC', '-o', str(so), str(src)])
  monkeypatch.setattr(_native, 'LIB_PATH', str(so))
  monkeypatch.setattr(_native, '_lib', None)
  with pytest.raises(_native.GeecoNativeError, match='ABI version mismatch'):
    _native.load()


def test_dev_switches_are_ignored_without_geeco_dev(monkeypatch):
  """GEECO_* A/B switches are development tooling: without GEECO_DEV=1 neither the Python side nor the library reads
  them (a stray variable in a production environment must not change which kernels run)."""
  from geeco_amd import _dev
  monkeypatch.delenv('GEECO_DEV', raising=False)
  monkeypatch.setenv('GEECO_NO_FUSED_BOTTOM', '1')
  assert not _dev.enabled() and _dev.env('GEECO_NO_FUSED_BOTTOM') is None and _dev.env('GEECO_WGRAD_STREAMS', '2') == '2'
  monkeypatch.setenv('GEECO_DEV', '1')
  assert _dev.enabled() and _dev.env('GEECO_NO_FUSED_BOTTOM') == '1'
  monkeypatch.setenv('GEECO_DEV', '0')
  assert not _dev.enabled()
  # the library: no getenv of a GEECO_ switch outside the one gate
  import glob
  for path in glob.glob(os.path.join(ROOT, 'geeco_amd', 'csrc', '*.hip')) + glob.glob(os.path.join(ROOT, 'geeco_amd', 'csrc', '*.cpp')):
    src = open(path).read()
    direct = re.findall(r'(?<![_a-z])getenv\("(GEECO_[A-Z0-9_]+)"', src)
    assert direct in ([], ['GEECO_DEV']), (path, direct)
  for path in glob.glob(os.path.join(ROOT, 'geeco_amd', '*.py')):
    if not path.endswith('_dev.py'):
      assert not re.findall(r"os\.environ\.get\('GEECO_", open(path).read()), path


def test_product_library_holds_no_switches_and_no_development_kernels():
  """the shipped library is the path that runs.  It reports itself as the product build, the name of no
  GEECO_* switch occurs in it (geeco_dev_getenv is the constant nullptr there: nothing reads the environment), and none of
  the kernels only a switch could select was compiled in (the development build, scripts/dev/build_dev_lib.sh, has them)."""
  from geeco_amd import _native
  lib = _native.load()
  assert lib.geeco_has_dev_kernels() == 0
  blob = open(os.path.join(ROOT, 'geeco_amd', 'libgeeco_hip.so'), 'rb').read()
  for name in re.findall(r'`(GEECO_[A-Z0-9_]+)', open(os.path.join(ROOT, 'scripts', 'dev', 'SWITCHES.md')).read()) + ['GEECO_DEV']:
    assert name.encode() not in blob, name
  for kernel in (b'conv1_conv2_fwd_kernel', b'conv_s2_halo_fwd_kernel', b'geeco_conv1_conv2_fwd'):
    assert kernel not in blob, kernel
  assert b'conv_s2_halo_fwd_ws_kernel' in blob
