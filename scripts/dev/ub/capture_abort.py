"""What HIP leaves behind when a stream capture fails (ROCm 7.2, gfx950) -- the question behind TrainStepRunner._capture's
fallback: a capture on s0 that forked work to s1 (event record / wait, as a process group's collective does with its own
stream) and then hit an operation that cannot be captured.  Prints return codes and capture states step by step, then tries
the ways out: EndCapture as is, joining s1 back and EndCapture, BeginCapture again on the same / a fresh stream.
  usage: python scripts/dev/ub/capture_abort.py            (raw HIP through ctypes; no torch involved)"""
import ctypes as C
hip = C.CDLL('libamdhip64.so')
hip.hipGetErrorName.restype = C.c_char_p
name = lambda rc: '%d %s' % (rc, hip.hipGetErrorName(rc).decode())
vp = C.c_void_p


def stream():
  s = vp()
  assert hip.hipStreamCreate(C.byref(s)) == 0
  return s


def status(s):
  st = C.c_int(-1)
  rc = hip.hipStreamIsCapturing(s, C.byref(st))
  return 'status=%s%s' % ({0: 'none', 1: 'active', 2: 'invalidated'}.get(st.value, st.value), '' if rc == 0 else ' (rc %s)' % name(rc))


def scenario(title, illegal, join_before_end):
  print('==== %s' % title)
  s0, s1 = stream(), stream()
  e, e2 = vp(), vp()
  hip.hipEventCreate(C.byref(e)); hip.hipEventCreate(C.byref(e2))
  buf = vp(); hip.hipMalloc(C.byref(buf), 1 << 20)
  print('begin(s0, thread_local):', name(hip.hipStreamBeginCapture(s0, 1)))
  print('  memset s0:', name(hip.hipMemsetAsync(buf, 0, 1024, s0)))
  print('  record e on s0:', name(hip.hipEventRecord(e, s0)), '| s1 waits e:', name(hip.hipStreamWaitEvent(s1, e, 0)))
  print('  memset s1:', name(hip.hipMemsetAsync(buf, 1, 1024, s1)), '|', 's0', status(s0), '| s1', status(s1))
  if illegal:
    print('  hipStreamSynchronize(s1) [not capturable]:', name(hip.hipStreamSynchronize(s1)), '|', 's0', status(s0), '| s1', status(s1))
    hip.hipGetLastError()
  if join_before_end:
    print('  join: record e2 on s1:', name(hip.hipEventRecord(e2, s1)), '| s0 waits e2:', name(hip.hipStreamWaitEvent(s0, e2, 0)))
  g = vp()
  print('end(s0):', name(hip.hipStreamEndCapture(s0, C.byref(g))), 'graph', g.value, '|', 's0', status(s0), '| s1', status(s1))
  hip.hipGetLastError()
  if not join_before_end:
    print('  now join: record e2 on s1:', name(hip.hipEventRecord(e2, s1)), '| s0 waits e2:', name(hip.hipStreamWaitEvent(s0, e2, 0)))
    g = vp()
    print('  end(s0) again:', name(hip.hipStreamEndCapture(s0, C.byref(g))), 'graph', g.value, '|', 's0', status(s0), '| s1', status(s1))
    hip.hipGetLastError()
  print('eager memset on s1:', name(hip.hipMemsetAsync(buf, 2, 1024, s1)), '| sync s1:', name(hip.hipStreamSynchronize(s1)))
  hip.hipGetLastError()
  print('eager memset on s0:', name(hip.hipMemsetAsync(buf, 2, 1024, s0)), '| sync s0:', name(hip.hipStreamSynchronize(s0)))
  hip.hipGetLastError()
  print('begin(s0) again:', name(hip.hipStreamBeginCapture(s0, 1)), '|', status(s0))
  g = vp()
  print('  memset + end:', name(hip.hipMemsetAsync(buf, 0, 1024, s0)), name(hip.hipStreamEndCapture(s0, C.byref(g))), 'graph', bool(g.value))
  hip.hipGetLastError()
  s2 = stream()
  print('fresh stream: begin', name(hip.hipStreamBeginCapture(s2, 1)), 'memset', name(hip.hipMemsetAsync(buf, 0, 1024, s2)),
        '| s1 forked into it again:', name(hip.hipEventRecord(e, s2)), name(hip.hipStreamWaitEvent(s1, e, 0)), name(hip.hipMemsetAsync(buf, 1, 1024, s1)),
        name(hip.hipEventRecord(e2, s1)), name(hip.hipStreamWaitEvent(s2, e2, 0)))
  g = vp()
  print('  end(s2):', name(hip.hipStreamEndCapture(s2, C.byref(g))), 'graph', bool(g.value), '| s1', status(s1))
  hip.hipGetLastError()
  print('device sync:', name(hip.hipDeviceSynchronize()))


def reset_scenario(title, empty):
  """The way out that worked above, looked at closely: BeginCapture on the SAME origin stream + a clean EndCapture; what state is
  the forked stream in afterwards, for eager work and for the next fork?"""
  print('==== %s' % title)
  s0, s1 = stream(), stream()
  e, e2 = vp(), vp()
  hip.hipEventCreate(C.byref(e)); hip.hipEventCreate(C.byref(e2))
  buf = vp(); hip.hipMalloc(C.byref(buf), 1 << 20)
  print('begin(s0):', name(hip.hipStreamBeginCapture(s0, 1)), '| fork s1:', name(hip.hipEventRecord(e, s0)), name(hip.hipStreamWaitEvent(s1, e, 0)),
        name(hip.hipMemsetAsync(buf, 1, 1024, s1)))
  g = vp()
  print('end(s0) unjoined:', name(hip.hipStreamEndCapture(s0, C.byref(g))), '| s0', status(s0), '| s1', status(s1))
  hip.hipGetLastError()
  print('reset: begin(s0):', name(hip.hipStreamBeginCapture(s0, 1)), '| s0', status(s0), '| s1', status(s1))
  if not empty:
    print('  memset s0:', name(hip.hipMemsetAsync(buf, 0, 1024, s0)))
  g = vp()
  print('reset: end(s0):', name(hip.hipStreamEndCapture(s0, C.byref(g))), 'graph', bool(g.value), '| s0', status(s0), '| s1', status(s1))
  hip.hipGetLastError()
  print('eager on s1: memset', name(hip.hipMemsetAsync(buf, 2, 1024, s1)), '| record e2', name(hip.hipEventRecord(e2, s1)), '| sync s1', name(hip.hipStreamSynchronize(s1)), '| s1', status(s1))
  hip.hipGetLastError()
  print('eager on s0: memset', name(hip.hipMemsetAsync(buf, 2, 1024, s0)), '| s0 waits e2', name(hip.hipStreamWaitEvent(s0, e2, 0)), '| sync s0', name(hip.hipStreamSynchronize(s0)), '| s0', status(s0))
  hip.hipGetLastError()
  s2 = stream()
  print('capture on a fresh stream with s1 forked and joined:', name(hip.hipStreamBeginCapture(s2, 1)), name(hip.hipEventRecord(e, s2)), name(hip.hipStreamWaitEvent(s1, e, 0)),
        name(hip.hipMemsetAsync(buf, 1, 1024, s1)), name(hip.hipEventRecord(e2, s1)), name(hip.hipStreamWaitEvent(s2, e2, 0)))
  g = vp()
  print('  end(s2):', name(hip.hipStreamEndCapture(s2, C.byref(g))), 'graph', bool(g.value), '| s1', status(s1))
  print('device sync:', name(hip.hipDeviceSynchronize()))


hip.hipSetDevice(0)
reset_scenario('unjoined fork -> reset with an empty capture', True)
reset_scenario('unjoined fork -> reset with a one-node capture', False)
scenario('unjoined fork, nothing illegal', False, False)
scenario('fork + illegal sync on the forked stream, end as is', True, False)
scenario('fork + illegal sync on the forked stream, joined before end', True, True)
