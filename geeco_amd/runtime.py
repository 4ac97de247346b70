"""Step runner: replays one training step as captured hipGraphs, with the data-parallel exchange.

The reference's hot loop is ``session.run(train_op)`` once per batch.  All shapes
are static, so the ~110 kernel launches of a step are captured once (HIP stream capture through
``torch.cuda.CUDAGraph``, our kernels are launched on torch's current stream) and replayed.

One GPU:      graph A (forward + backward)  ->  graph B (Adam + re-derived weight copies).
Data parallel (buckets in reverse-layer order, overlapped with the backward):
  graph A1 = forward + decoder backward + encoder backward down to conv3 (every gradient except the
             encoders' conv1 / conv2 is final: 99.4 % of the bytes)
  -> EARLY bucket: ONE asynchronous RCCL all-reduce over the span of those arena ranges on the communicator's stream
  graph A2 = conv2's filter gradient + the fused encoder bottom (conv2 dgrad + conv1 wgrad, ~0.9 ms), which write
             the conv1 / conv2 gradients straight into one small staging buffer (nothing touches the arena meanwhile)
  -> LATE bucket: all-reduce of the staging buffer (177 KB)
  graph B  = wait for both, unpack the staging buffer, Adam (grad_scale = 1 / world).
The early bucket therefore runs beside A2's kernels; only the small late bucket is exposed.
The Adam step counter and lr_t live in device memory, so replays advance them correctly.
"""
from __future__ import annotations

import re
import threading

import torch

from . import _dev
from . import dist as gdist

# hipGraph capture in the default ("global") mode is invalidated by a hipMalloc / synchronous copy issued by ANOTHER
# thread during the capture window (the input prefetcher uploads episodes in a background thread): both sides take
# this lock (geeco_amd/input_fn.py: episode_to_device).
CAPTURE_LOCK = threading.RLock()
# 'thread_local': HIP calls of OTHER threads (the RCCL watchdog polling its events, a prefetch upload that slipped past
# the lock) neither fail nor invalidate the capture; only this thread's stream work is recorded.
_CAPTURE_MODE = 'thread_local'

_LATE = re.compile(r'/conv[12]/(kernel|bias)$')


def gradient_buckets(store):
  """(early ranges, late ranges) of the flat gradient arena as (offset, length) in floats.  late = the encoders'
  conv1 / conv2 variables (their gradients come out of the last two launches of the backward); early = the rest,
  merged into maximal contiguous ranges (alignment pads included)."""
  late = []
  names = list(store.shapes.keys())
  for i, n in enumerate(names):
    if _LATE.search(n):
      lo = store.offsets[n]
      hi = store.offsets[names[i + 1]] if i + 1 < len(names) else store.size
      if late and late[-1][1] == lo:
        late[-1][1] = hi
      else:
        late.append([lo, hi])
  early, pos = [], 0
  for lo, hi in late:
    if lo > pos:
      early.append((pos, lo - pos))
    pos = hi
  if store.size > pos:
    early.append((pos, store.size - pos))
  return early, [(lo, hi - lo) for lo, hi in late]


class CaptureFailed(RuntimeError):
  """A one-graph capture of the data-parallel step failed; the process cannot use the streams involved again (TrainStepRunner._capture)."""


class TrainStepRunner:
  """``dp``: run the three-part data-parallel step (None = when the process group has more than one rank; True forces it
  with one rank, so that the exchange really goes through the backend -- tests).  ``overlap``: the early bucket goes out
  beside part 2 (default); False = both buckets after part 2 (``bench.py --dp-serial``: the difference between the two
  is what the overlap buys on a given node)."""

  def __init__(self, model, use_graph=True, warmup=2, dp=None, overlap=True, reserved_cus=0, capture_exchange=None, eager_adam=False):
    """``reserved_cus``: CUs the two persistent kernels of part 2 leave to the collective that runs beside them (0 = none;
    a launch argument of those kernels, applied around THIS runner's part 2 only).  ``eager_adam`` (three-graph form only): part 3 is
    not a graph but the optimiser's two pieces launched eagerly, the early one as soon as the early bucket has arrived, beside the late
    bucket's all-reduce (two graph launches per step).  ``capture_exchange``: capture the
    whole data-parallel step -- the three parts AND both all-reduces, the early one as a branch beside part 2 -- into ONE
    hipGraph (RCCL's launches are stream work like any other; the fork to the communicator's stream and the joins become
    graph edges): one graph launch per step instead of three and no host-side stream joins (measured at one rank, bench.py
    ``dp_one_rank``: +18 us over the single-GPU step instead of +71).  ``overlap`` / ``skip_allreduce`` are then fixed at
    capture time; a backend whose collectives cannot be captured (anything but RCCL) gets the three-graph form, told once; a
    capture that FAILS raises CaptureFailed and ends the run (``_capture`` says why there is no way back).
    None / False (default) = three graphs with the exchange launched eagerly between them: the form that is safe by
    construction (every collective is an ordinary RCCL launch, so a rank that replays and a rank that runs eagerly -- ragged
    end of an epoch, ``null_step`` -- issue the same kind of call on the communicator).  The one-graph form has only ever run
    with ONE rank on hardware (no multi-GPU box was available to any round), so it is opt-in: ``dp_form`` of RunConfig /
    ``--dp_form`` of scripts/train_e2evmc.py, and bench.py, which times the safe form first and the captured forms behind it."""
    self.model = model
    self.capture_exchange = bool(capture_exchange)
    self.eager_adam = bool(eager_adam)
    self.reserved_cus = int(reserved_cus)
    self.world = gdist.world_size()
    model.world = self.world
    self.dp = (self.world > 1) if dp is None else bool(dp)
    self.overlap = bool(overlap)
    self.use_graph = bool(use_graph) and torch.cuda.is_available()
    self._graphs = None
    self._warm = warmup
    self._calls = 0
    self.skip_allreduce = False          # bench.py: measure the step without the exchange
    self.early, self.late = gradient_buckets(model.store)
    self.staging = None
    self.redirected = False
    if self.dp and self.late:
      store = model.store
      # one staging buffer per variable store: models built for other batch sizes (ragged last batch) share it
      if getattr(store, 'late_staging', None) is None:
        store.late_staging = torch.zeros(sum(n for _, n in self.late), dtype=torch.float32, device=store.grads.device)
      self.staging = store.late_staging
      # The bottom of the backward writes the conv1 / conv2 gradients STRAIGHT into the staging buffer (no pack copies),
      # so nothing touches the arena while the early bucket is in flight and the early bucket can be ONE all-reduce
      # over the span of its ranges: the late variables' slots inside that span are dead (zero) until part 3 unpacks.
      # The redirection is OWNED by this runner: it is switched on around this runner's own part 2 only (eager run or capture;
      # the pointers are baked into the captured graph) and off again right behind it, so no model-wide state outlives a
      # call -- a second runner on the same model (another batch size, a non-dp probe) neither inherits nor undoes it.
      # Between part 2 and part 3 ``store.grads`` is NOT authoritative for conv1 / conv2: their gradients sit in the
      # staging buffer until part 3 has unpacked them (nothing reads those slots in between).
      redirect = getattr(model, 'redirect_late_gradients', None)
      self.redirected = bool(redirect and redirect(self.staging, self.late))
      if self.redirected:
        redirect(None, None)
    # The optimiser step in two pieces (whole data-parallel step in one pass over the stream: eager, or captured as ONE graph):
    # everything that came with the early bucket is updated while the late bucket is on the wire, the late bucket's variables
    # follow with their gradients read straight from the staging buffer (no unpack copies).  The three-graph form runs the same two
    # pieces back to back as its part 3 (round 6; it kept the unpack copies + one whole Adam launch before).
    self.split_adam = (self.dp and self.staging is not None and hasattr(model, 'apply_gradients_of')
                       and 1 <= len(self.early) <= 8 and 1 <= len(self.late) <= 8 and _dev.env('GEECO_NO_SPLIT_ADAM') is None)
    # Single GPU: the optimiser's streaming work (slab sums of conv3..conv8, Adam over 99 % of the arena) runs on a second stream
    # beside the fused encoder-bottom backward instead of behind it (graph._ModelBase.backward_and_apply; bitwise the plain step)
    self.beside_bottom = (not self.dp and hasattr(model, 'backward_and_apply') and model.can_apply_beside_bottom()
                          and 1 <= len(self.early) <= 8 and 1 <= len(self.late) <= 8 and _dev.env('GEECO_NO_ADAM_BESIDE_BOTTOM') is None)
    # ... and the ONE-GRAPH forms of the data-parallel step (and their eager warm-up steps) do the same with Adam's early piece.  RCCL only:
    # with gloo the form is three graphs anyway, and two gloo ranks that share one GPU (the rehearsals of tests/) run 10 x slower once
    # every process drives one more hardware queue -- the side stream is otherwise only ever used inside captures.
    self.dp_beside_bottom = (self.dp and self.overlap and self.capture_exchange and self.split_adam and self.redirected
                             and hasattr(model, 'can_apply_beside_bottom') and model.can_apply_beside_bottom() and gdist.backend() == 'nccl'
                             and _dev.env('GEECO_NO_DP_ADAM_BESIDE') is None)
    if self.redirected and self.early:
      lo = min(off for off, _ in self.early)
      hi = max(off + n for off, n in self.early)
      self.early_calls = [(lo, hi - lo)]
    else:
      self.early_calls = list(self.early)

  def bucket_info(self):
    return {'graphs_per_step': (1 if (self.capture_exchange or not self.dp) else 2 if (self.eager_adam and self.split_adam) else 3) if self.use_graph else 0,
            'early_bytes': 4 * sum(n for _, n in self.early), 'early_ranges': len(self.early),
            'early_allreduce_calls': len(self.early_calls), 'early_bytes_on_the_wire': 4 * sum(n for _, n in self.early_calls),
            'late_bytes': 4 * sum(n for _, n in self.late), 'late_ranges': len(self.late),
            'late_written_in_place': self.redirected, 'mode': 'overlap' if self.overlap else 'serial'}

  # -- pieces of a step ----------------------------------------------------------------------------------
  def _part1(self):
    self.model.forward(backward_too=True)
    if self.dp:
      self.model.backward(part='upper')
    else:
      self.model.backward(adam_prepare=True)      # (the optimiser's scalars ride in the backward's last slab-sum launch)

  def _part2(self, prepare=True, before_bottom=None):
    enc = getattr(self.model, 'enc', None)
    if self.redirected:
      self.model.redirect_late_gradients(self.staging, self.late)
    if enc is not None and self.reserved_cus:
      enc.reserved_cus = self.reserved_cus
    try:
      if before_bottom is None:
        self.model.backward(part='bottom', adam_prepare=prepare)
      else:
        self.model.backward(part='bottom', adam_prepare=prepare, before_bottom=before_bottom)
    finally:
      if enc is not None:
        enc.reserved_cus = 0
      if self.redirected:
        self.model.redirect_late_gradients(None, None)
    if self.redirected or self.staging is None:
      return
    g, pos = self.model.store.grads, 0
    for off, n in self.late:
      self.staging[pos:pos + n].copy_(g[off:off + n])
      pos += n

  def _part3_early(self):
    g = self.model.store.grads
    self.model.apply_gradients_of([(g[off:off + n], off, n) for off, n in self.early], last=False)

  def _part3_late(self):
    segs, pos = [], 0
    for off, n in self.late:
      segs.append((self.staging[pos:pos + n], off, n))
      pos += n
    self.model.apply_gradients_of(segs, g_out=self.model.store.grads, last=True)     # (the gradient arena gets them too)

  def _part3(self):
    if self.dp and self.staging is not None:
      g, pos = self.model.store.grads, 0
      for off, n in self.late:
        g[off:off + n].copy_(self.staging[pos:pos + n])
        pos += n
    self.model.apply_gradients()

  def _whole_step(self):
    if self.beside_bottom:
      self.model.forward(backward_too=True)
      self.model.backward_and_apply(self.early, self.late)
      return
    self._part1()
    self._part3()

  def _exchange_early(self):
    if self.skip_allreduce:
      return []
    g = self.model.store.grads
    return [gdist.allreduce_async(g[off:off + n]) for off, n in self.early_calls]

  def _exchange_late(self):
    if self.skip_allreduce or self.staging is None:
      return []
    return [gdist.allreduce_async(self.staging)]

  def _dp_step_beside(self):
    """The one-pass data-parallel step (eager, or under ONE capture) with the early piece of the optimiser step on a second stream
    beside the fused encoder-bottom backward, as the single-GPU step has it (graph._ModelBase.backward_and_apply, finding 38):
    the early bucket's all-reduce runs beside conv3's input gradient and conv2's filter gradient and has normally arrived when
    the fused bottom starts; Adam's pass over it then streams beside the bottom instead of behind the late bucket's launch.  The
    optimiser's scalars ride in the slab-sum launch of part 1 (they must exist before the early piece).  If the bucket is late the
    early piece simply runs when it arrives -- behind the bottom at worst, where it used to be."""
    m = self.model
    main = torch.cuda.current_stream()
    side = m.optimizer_stream()
    m.forward(backward_too=True)
    m.backward(part='upper', adam_prepare=True)
    works = self._exchange_early()
    ev = torch.cuda.Event()
    marked = []

    def mark(pending=None):
      ev.record(main)
      marked.append(True)
    self._part2(prepare=False, before_bottom=mark)
    if not marked:
      mark()
    late = self._exchange_late()           # (issued before the side work: its launch is on the critical path, the early piece is not)
    side.wait_event(ev)                    # behind conv3's input gradient (reads conv3's kernel) and behind the bottom's launch packet
    with torch.cuda.stream(side):
      for w in works:
        w.wait()
      self._part3_early()
    for w in late:
      w.wait()
    main.wait_stream(side)
    self._part3_late()

  def _dp_step(self, run=None):
    """The data-parallel step: ``run`` = the three parts as callables (captured graphs' replays or the eager functions)."""
    whole = run is None                    # not three replayed graphs: the parts run right here (eagerly, or under ONE capture)
    if whole and self.dp_beside_bottom:
      return self._dp_step_beside()
    run = run or [self._part1, self._part2, self._part3]
    run[0]()
    works = self._exchange_early() if self.overlap else []   # on the communicator's stream, behind part 1, beside part 2
    run[1]()
    if not self.overlap:
      works = self._exchange_early()
    late = self._exchange_late()
    if whole and self.split_adam:
      for w in works:
        w.wait()                           # the compute stream waits; the host does not (RCCL)
      self._part3_early()                  # 99 % of the update, beside the late bucket's all-reduce
      for w in late:
        w.wait()
      self._part3_late()
      return
    if len(run) == 2:                      # two graphs + the optimiser's pieces launched eagerly (eager_adam)
      # part 2's graph carries the optimiser's per-step scalars (adam_prepare rides in its last slab-sum launch); a REPLAY runs no
      # python, so the flag that tells apply_gradients_of not to prepare again is set here
      self.model._prepared = True
      for w in works:
        w.wait()
      self._part3_early()                  # beside the late bucket's all-reduce
      for w in late:
        w.wait()
      self._part3_late()
      return
    for w in works + late:
      w.wait()
    run[2]()

  def _part3_pieces(self):
    """Part 3 of the three-graph form: the optimiser step as the same two pieces the one-pass forms use, back to back (both buckets have
    arrived by then) -- conv1 / conv2 are updated from the staging buffer, so the three unpack copies of ``_part3`` are not launched."""
    self._part3_early()
    self._part3_late()

  def _parts(self):
    if not self.dp:
      return [self._whole_step]
    if self.capture_exchange:
      return [self._dp_step]
    if self.eager_adam and self.split_adam:
      return [self._part1, self._part2]
    return [self._part1, self._part2, self._part3_pieces if self.split_adam else self._part3]

  def _capture(self):
    # the warm-up steps before this call already ran eagerly
    # single process: the whole step is one graph (no inter-graph launch gap); data parallel: the exchange sits between
    def capture(parts):
      graphs = []
      with CAPTURE_LOCK:
        for fn in parts:
          g = torch.cuda.CUDAGraph()
          with torch.cuda.graph(g, capture_error_mode=_CAPTURE_MODE):
            fn()
          graphs.append(g.replay)
      return graphs
    if self.dp and self.capture_exchange and gdist.group_active() and gdist.backend() != 'nccl':
      # Decided BEFORE anything is captured: only RCCL's launches are stream work a hipGraph can hold; any other backend's
      # collective on a device tensor (gloo: copies through the host, stream synchronisation) cannot be captured, and an attempt
      # cannot be undone (below).  Such a group gets the three-graph form, told once.
      import warnings
      warnings.warn("TrainStepRunner: capture_exchange asked for, but the '%s' backend's collectives cannot be captured into a "
                    'hipGraph; using three graphs with the exchange between them' % gdist.backend())
      self.capture_exchange = False
    if self.dp and self.capture_exchange:
      # No in-process fallback from a FAILED attempt, by measurement (scripts/dev/ub/capture_abort.py, ROCm 7.2,
      # profiles/r06/ub_capture_abort.txt): when a capture breaks with a forked stream not joined back -- the shape every failure
      # inside a collective has, the communicator's stream being the fork -- hipStreamEndCapture returns an error and leaves the
      # origin AND the forked stream in capture mode; neither a second EndCapture, nor joining the fork afterwards, nor
      # BeginCapture on the same stream brings them back, and eager work on either stream then fails.  (First build of this
      # round tried: capture_end raised `capturing stream has unjoined work`, the stream stayed in capture mode.)  So a failed
      # one-graph capture ENDS the run: the current stream is put back (torch.cuda.graph leaves its side stream current when
      # capture_end raises), and CaptureFailed goes up -- bench.py answers with the safe form's line it measured first, a
      # training script exits non-zero and is restarted with --dp_form three_graphs.  Nothing was executed by the attempt.
      orig = torch.cuda.current_stream()
      try:
        self._graphs = capture(self._parts())
      except RuntimeError as e:
        torch.cuda.set_stream(orig)
        self.model._prepared = False
        raise CaptureFailed('capturing the data-parallel step (with both all-reduces) into one hipGraph failed: %s: %s.  HIP '
                            'leaves the streams of a failed capture in capture mode, so this process cannot capture or launch on '
                            'them again: restart with a three_graphs* form (three_graphs_reserve16 is the default)' % (type(e).__name__, str(e)[:300])) from e
      return
    self._graphs = capture(self._parts())

  def prepare(self):
    """Untimed set-up for benchmarks: run the eager warm-up steps and capture the graphs now, so that
    no later call pays for the capture."""
    while self.use_graph and self._graphs is None:
      self.step()

  def step(self):
    """One optimiser step on the batch currently in ``model.inputs``."""
    if self.use_graph and self._graphs is None and self._calls >= self._warm:
      # NB capture itself does not execute the step; fall through to replay
      self._capture()
    self._calls += 1
    if self._graphs is None:               # eager (warm-up steps, use_graph=False)
      if self.dp:
        self._dp_step()
      else:
        self._whole_step()
    elif len(self._graphs) == 1:
      self._graphs[0]()
    else:
      self._dp_step(self._graphs)

  def null_step(self):
    """A step of a rank that holds no sample (ragged end of an epoch under data parallelism): zero gradients into
    the exchange, then the same Adam update as every other rank."""
    self._calls += 1
    self.model.store.grads.zero_()
    if self.staging is not None:
      self.staging.zero_()
    for w in self._exchange_early() + self._exchange_late():
      w.wait()
    self._part3()


# Every form of the data-parallel step by name (RunConfig.dp_form, --dp_form, bench.py's comm.step_ms keys).
DP_FORMS = {
    'three_graphs_reserve16': dict(overlap=True, capture_exchange=False, reserved_cus=16),   # DEFAULT: exchange launched eagerly between three
                                                                         # graphs; part 2's persistent kernels leave 16 CUs to RCCL (finding 37)
    'three_graphs': dict(overlap=True, capture_exchange=False),          # ... every CU to part 2 (RCCL's workgroups then wait for CUs or take them)
    'three_graphs_serial': dict(overlap=False, capture_exchange=False),  # ... both buckets behind part 2
    'two_graphs': dict(overlap=True, capture_exchange=False, eager_adam=True),   # parts 1 and 2 as graphs, the optimiser's two pieces launched
                                                                                 # eagerly: the early one beside the late bucket's all-reduce (as safe as
                                                                                 # three_graphs: every collective an ordinary launch; -8 us at one rank)
    'two_graphs_reserve16': dict(overlap=True, capture_exchange=False, eager_adam=True, reserved_cus=16),   # ... part 2 leaves 16 CUs to RCCL
    'two_graphs_reserve32': dict(overlap=True, capture_exchange=False, eager_adam=True, reserved_cus=32),   # ... or 32 (should RCCL open more channels)
    'two_graphs_serial': dict(overlap=False, capture_exchange=False, eager_adam=True),                      # ... both buckets behind part 2
    'overlap': dict(overlap=True, capture_exchange=True),                # ONE graph, early bucket beside part 2, every CU to compute
    'overlap_reserve16': dict(overlap=True, capture_exchange=True, reserved_cus=16),   # ... part 2's persistent kernels leave 16 CUs to RCCL
    'overlap_reserve32': dict(overlap=True, capture_exchange=True, reserved_cus=32),   # ... or 32 (RCCL's channel count decides)
    'serial': dict(overlap=False, capture_exchange=True),                # ONE graph, both buckets behind part 2
}
DP_FORM_DEFAULT = 'three_graphs_reserve16'
# bench.py's trials: first the forms in which every collective is an ordinary launch (whether the early bucket should run beside part 2 at
# all -- part 2's kernels fill every CU's registers -- with CUs left to RCCL, or behind it, is a property of the node), then the captured ones
DP_CANDIDATES_SAFE = tuple((k, DP_FORMS[k]) for k in ('three_graphs', 'two_graphs', 'two_graphs_reserve16', 'two_graphs_reserve32', 'two_graphs_serial',
                                                      'three_graphs_serial'))
DP_CANDIDATES_CAPTURED = tuple((k, DP_FORMS[k]) for k in ('overlap', 'overlap_reserve16', 'overlap_reserve32', 'serial'))
DP_CANDIDATES = DP_CANDIDATES_SAFE + DP_CANDIDATES_CAPTURED


def dp_form_kwargs(name=None):
  """TrainStepRunner arguments of a named form.  None = the safe default (the product reads no switch from the environment)."""
  name = name or DP_FORM_DEFAULT
  if name not in DP_FORMS:
    raise ValueError("unknown dp_form '%s' (one of %s)" % (name, ', '.join(DP_FORMS)))
  return dict(DP_FORMS[name])


def pick_dp_runner(model, use_graph=True, steps=8, candidates=DP_CANDIDATES, log=None):
  """NB every candidate runs ~13 REAL optimiser steps on the model's live parameters (fine for bench.py's repeated batch;
  anything that cares about its trajectory passes ``dp_form`` instead).  N > 1: which form of the exchange is fastest ON THIS NODE is not knowable in advance -- the early bucket runs beside two
  persistent one-block-per-CU kernels (0.83 ms), and whether RCCL's workgroups slow those blocks down by more than the
  collective hides depends on the link topology and RCCL's channel count.  So measure: every candidate captures its graph(s)
  and runs ``steps`` real optimiser steps between barriers; the time is the MAX over ranks (identical on every rank, so all
  ranks pick the same candidate without another exchange).  Returns (runner of the fastest form, {name: ms per step}).  With one
  rank (or one candidate) nothing is measured."""
  import time
  world = gdist.world_size()
  if world == 1 or len(candidates) == 1:
    name, kw = candidates[0]
    return TrainStepRunner(model, use_graph=use_graph, warmup=2, **kw), {}
  dev = model.store.params.device
  timings, runners = {}, {}
  for name, kw in candidates:
    r = TrainStepRunner(model, use_graph=use_graph, warmup=2, **kw)
    r.prepare()
    for _ in range(3):
      r.step()
    torch.cuda.synchronize()
    torch.distributed.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
      r.step()
    torch.cuda.synchronize()
    timings[name] = gdist.max_over_ranks((time.perf_counter() - t0) / steps * 1e3, dev)
    runners[name] = r
    if log:
      log('dp form %-18s %.4f ms/step (max over %d ranks, %d steps, %d graph(s) per step)' %
          (name, timings[name], world, steps, r.bucket_info()['graphs_per_step']))
  best = min(timings, key=lambda k: (timings[k], list(timings).index(k)))
  return runners[best], timings


class EvalStepRunner:
  """Forward-only replay (Estimator.evaluate / predict)."""

  def __init__(self, model, use_graph=True, warmup=1):
    self.model = model
    self.use_graph = bool(use_graph) and torch.cuda.is_available()
    self._g = None
    self._warm = warmup
    self._calls = 0

  def step(self):
    if self.use_graph and self._g is None and self._calls >= self._warm:
      g = torch.cuda.CUDAGraph()
      with CAPTURE_LOCK:
        with torch.cuda.graph(g, capture_error_mode=_CAPTURE_MODE):
          self.model.forward(backward_too=False)
      self._g = g
    self._calls += 1
    if self._g is not None:
      self._g.replay()
    else:
      self.model.forward(backward_too=False)
