"""Known-answer tests pinning the CPU oracle: analytic facts derivable from the
reference source + the TF-1.15 op semantics it relies on.  The reference itself has no tests, so
these (and the self-golden fixtures) are all that pins the oracle: PARITY UNPINNED."""
import math

import numpy as np
import torch

from oracle import geeco_oracle as O


def test_alpha_tables():
  np.testing.assert_allclose(O.dynimg_alpha(2), [-0.5, 0.5], atol=1e-7)
  np.testing.assert_allclose(O.dynimg_alpha(4), [-2.416667, 0.583333, 1.083333, 0.75], atol=2e-6)
  a16 = [-25.472393, -10.472393, -3.972393, -0.305726, 1.944274, 3.344274, 4.177607, 4.606179, 4.731179, 4.620067,
         4.320067, 3.865522, 3.282189, 2.589881, 1.804167, 0.9375]
  np.testing.assert_allclose(O.dynimg_alpha(16), a16, atol=1e-5)
  for T in (2, 3, 4, 8, 16, 32):
    a = O.dynimg_alpha(T).astype(np.float64)
    assert abs(float(a.sum())) < 1e-6 * T * (T + 1)                          # sum alpha = 0 up to float32 rounding


def test_dynimg_known_answers():
  const = torch.full((2, 5, 4, 4, 3), 0.3)
  assert float(O.dynimg(const.double()).abs().max()) < 1e-9                   # constant sequence -> 0
  cur = torch.rand(1, 6, 6, 3, dtype=torch.float64)
  assert float(O.dynimg(torch.stack([cur, cur], 1)).abs().max()) == 0.0      # dyndiff(cur == tgt) == 0
  tgt = torch.rand(1, 6, 6, 3, dtype=torch.float64)
  d = 0.5 * (tgt - cur)
  ref = (d - d.min()) / (d.max() - d.min() + 1e-6)
  torch.testing.assert_close(O.dynimg(torch.stack([cur, tgt], 1)), ref)


def test_same_padding_alignment_probe():
  # even input, stride 2: output o covers inputs {2o, 2o+1, 2o+2}; delta at (1,1) lights ONLY output (0,0)
  assert O.same_pad(256, 3, 2) == (128, 0, 1)
  assert O.same_pad(256, 3, 1) == (256, 1, 1)
  assert O.same_pad(5, 3, 2) == (3, 1, 1)
  w = torch.zeros(3, 3, 1, 1, dtype=torch.float64)
  w[:, :, 0, 0] = 1.0
  for pos, expect in (((0, 0), [(0, 0)]), ((1, 1), [(0, 0)]), ((2, 2), [(0, 0), (0, 1), (1, 0), (1, 1)])):
    x = torch.zeros(1, 8, 8, 1, dtype=torch.float64)
    x[0, pos[0], pos[1], 0] = 1.0
    y = O.conv2d_same(x, w, torch.zeros(1, dtype=torch.float64), 2, relu=False)[0, :, :, 0]
    lit = sorted((int(i), int(j)) for i, j in torch.nonzero(y))
    assert lit == expect, (pos, lit)


def test_conv_torch_path_matches_independent_numpy():
  r = np.random.default_rng(0)
  for (H, W, Ci, Co, s) in ((8, 8, 3, 5, 1), (8, 10, 4, 6, 2), (7, 5, 2, 3, 2)):
    x = r.standard_normal([2, H, W, Ci]); w = r.standard_normal([3, 3, Ci, Co]); b = r.standard_normal([Co])
    a = O.conv2d_same(torch.tensor(x), torch.tensor(w), torch.tensor(b), s).numpy()
    np.testing.assert_allclose(a, O.conv2d_same_numpy(x, w, b, s), rtol=1e-10, atol=1e-10)


def test_parameter_counts():
  enc = O.count_parameters(O.encoder_param_shapes('x', 3, 256))
  assert enc == 1960496
  f = O.make_config(proc_obs='dynimg', proc_tgt='dyndiff')
  assert O.count_parameters(O.model_param_shapes(f, True)) == 7552796
  assert O.count_parameters(O.model_param_shapes(O.make_config(), False)) == 2583228
  s = O.model_param_shapes(f, True)
  assert s['GoalVMC/LSTMDecoder/lstm_cell/kernel'] == (3228, 512)
  assert s['GoalVMC/DynDiffEncoder/conv2/kernel'] == (3, 3, 32, 48)


def test_concat_layouts():
  obs = torch.arange(2 * 4 * 3, dtype=torch.float64).reshape(2, 2, 2, 3)
  dyn, tgt = obs + 100, obs + 200
  jnt = torch.tensor([[7., 8.], [9., 10.]], dtype=torch.float64)
  v2 = O.representation_concatenation_v2(obs, dyn, jnt, tgt)
  assert v2.shape == (2, 4 * 11)
  cell1 = v2[0, 11:22]       # cell (h=0,w=1): [obs(3) | dyn(3) | jnt(2) | tgt(3)]
  assert cell1.tolist() == [3, 4, 5, 103, 104, 105, 7, 8, 203, 204, 205]
  v1 = O.representation_concatenation(obs, tgt, jnt)
  assert v1[0, :8].tolist() == [0, 1, 2, 7, 8, 200, 201, 202]                 # jnt in the MIDDLE
  sc = O.state_concatenation(obs, jnt)
  assert sc[1, :5].tolist() == [12, 13, 14, 9, 10]


def test_lstm_zero_weights():
  N, D, H = 3, 5, 4
  c0 = torch.randn(N, H, dtype=torch.float64)
  c, h = O.lstm_cell(torch.randn(N, D, dtype=torch.float64), c0, torch.zeros(N, H, dtype=torch.float64),
                     torch.zeros(D + H, 4 * H, dtype=torch.float64), torch.zeros(4 * H, dtype=torch.float64))
  sig1 = 1 / (1 + math.exp(-1.0))
  torch.testing.assert_close(c, sig1 * c0)                                    # c' = sigmoid(0 + forget_bias) c
  torch.testing.assert_close(h, 0.5 * torch.tanh(c))                          # h' = sigmoid(0) tanh(c')


def test_loss_known_answers():
  N = 5
  logits = torch.zeros(N, 3, dtype=torch.float64)
  assert abs(float(O.softmax_xent(logits, torch.tensor([0, 1, 2, 1, 0]), 3)) - math.log(3.0)) < 1e-12   # ln 3
  t = torch.randn(N, 3, dtype=torch.float64)
  assert abs(float(O.mse(torch.zeros_like(t), t)) - float((t ** 2).mean())) < 1e-15
  cmd = torch.tensor([[0., 0., 0., -1.], [0., 0., 0., 0.49], [0., 0., 0., 0.51], [0., 0., 0., 1.]])
  tg = O.build_targets({'ee_state': torch.zeros(4, 2, 7), 'obj_state': torch.zeros(4, 2, 7)}, {'cmd': cmd}, O.make_config())
  assert tg['cmd_grp'].tolist() == [0, 1, 2, 2]                               # rint + 1
  assert torch.round(torch.tensor([0.5, 1.5, -0.5])).tolist() == [0.0, 2.0, -0.0]   # half-to-even like tf.math.rint


def test_adam_step_one():
  lr = 1e-4
  g = np.array([1e-3, -2.0, 3e-7, 0.0])
  p, m, v = np.zeros(4), np.zeros(4), np.zeros(4)
  O.adam_step_tf(p, g, m, v, 1, lr)
  expect = -lr * g / (np.abs(g) + 1e-8 / math.sqrt(1 - 0.999))               # = -lr g / (|g| + 3.1623e-7)
  np.testing.assert_allclose(p, expect, rtol=1e-9, atol=1e-18)
  assert abs(1e-8 / math.sqrt(1 - 0.999) - 3.1623e-7) < 1e-10


def test_error_surface():
  import pytest
  with pytest.raises(ValueError):
    O.model_param_shapes(O.make_config(proc_obs='nope'), True)
  with pytest.raises(ValueError):
    O.model_param_shapes(O.make_config(proc_tgt='nope'), True)
  with pytest.raises(ValueError):
    O.decoder_param_shapes('s', 10, O.make_config(control_mode='nope'))
  assert O.make_config(bogus_key=3) == O.make_config()                        # unknown keys silently dropped


def test_chunked_evaluation_matches_plain():
  """loss_and_grads_chunked (used by the full-size GPU parity tests) is the same mathematics as
  loss_and_grads: loss, predictions and every gradient agree in fp64 for both model kinds."""
  for goal, kw, N in ((True, dict(proc_obs='dynimg', proc_tgt='dyndiff', window_size=3), 3),
                      (True, dict(proc_obs='dynimg', proc_tgt='dyndiff', window_size=2, img_channels=4), 2),
                      (False, dict(window_size=3), 3)):
    cfg = O.make_config(img_height=136, img_width=136, batch_size=N, **kw)
    P = O.init_params(O.model_param_shapes(cfg, goal), seed=3)
    r = np.random.default_rng(4)
    for k in P:
      if k.endswith('/bias'):
        P[k] = (0.05 * r.standard_normal(P[k].shape)).astype(np.float32)
    feats, labels = O.synthetic_batch(cfg, goal, N, seed=5, H=136, W=136)
    tr = O.OracleTrainer(cfg, goal, P, dtype=torch.float64)
    loss, parts, grads, pred, _ = tr.loss_and_grads(feats, labels)
    loss2, parts2, grads2, pred2, ep2 = O.loss_and_grads_chunked(tr, feats, labels, chunk=2)
    assert abs(float(loss) - float(loss2)) < 1e-12 * abs(float(loss))
    for k in pred:
      torch.testing.assert_close(pred2[k], pred[k], rtol=1e-11, atol=1e-13)
    for k in grads:
      torch.testing.assert_close(grads2[k], grads[k], rtol=1e-9, atol=1e-14, msg=k)
    assert set(grads2) == set(grads)
    assert len(ep2['conv8_first_last']) == (3 if goal else 1)


def test_chunked_mask_consistent_mode():
  """The mask-consistent mode of loss_and_grads_chunked (full-size GPU parity): fed the restatement's OWN inputs and ReLU
  decisions it reproduces the plain gradients and reports zero disagreements; one decision flipped (at an entry whose
  pre-activation is far from zero) is counted with its |z| and moves the gradients -- i.e. the count and the bound on
  |z| the GPU tests assert really see a wrong mask."""
  cfg = O.make_config(img_height=136, img_width=136, batch_size=2, proc_obs='dynimg', proc_tgt='dyndiff', window_size=2)
  P = O.init_params(O.model_param_shapes(cfg, True), seed=3)
  feats, labels = O.synthetic_batch(cfg, True, 2, seed=5, H=136, W=136)
  tr = O.OracleTrainer(cfg, True, P, dtype=torch.float64)
  loss, _, grads, _, _ = O.loss_and_grads_chunked(tr, feats, labels, chunk=1)
  f = tr._cast(feats)
  jobs, _ = O._encoder_jobs(f, cfg, True, torch.float64)
  Pe = {k: v for k, v in tr.P.items()}
  acts = []
  for scope, x in jobs:
    col = {}
    O.conv_encoder(x, Pe, scope, col)
    acts.append([col['%s/conv%d' % (scope, i + 1)] > 0 for i in range(8)])
  mf = lambda j, i0, i1: [m[i0:i1] for m in acts[j]]
  loss2, _, grads2, _, ep2 = O.loss_and_grads_chunked(tr, feats, labels, chunk=1, encoder_inputs=[x for _, x in jobs], masks_fn=mf)
  assert float(loss2) == float(loss)
  for k in grads:
    torch.testing.assert_close(grads2[k], grads[k], rtol=1e-12, atol=1e-15, msg=k)
  for scope, st in ep2['relu_disagreements'].items():
    assert all(n == 0 and z == 0.0 and tot > 0 for n, z, tot in st), (scope, st)
  # flip one decision of conv2 in the second encoder, frame 1
  idx = (1, 5, 7, 3)
  acts[1][1][idx] = ~acts[1][1][idx]
  _, _, grads3, _, ep3 = O.loss_and_grads_chunked(tr, feats, labels, chunk=1, encoder_inputs=[x for _, x in jobs], masks_fn=mf)
  st = ep3['relu_disagreements']['GoalVMC/DynBuffEncoder']
  assert st[1][0] == 1 and st[1][1] > 1e-6 and sum(s[0] for s in st) == 1
  assert float(ep3['relu_disagreements']['GoalVMC/ConvEncoder'][1][0]) == 0
  assert not torch.equal(grads3['GoalVMC/DynBuffEncoder/conv2/kernel'], grads['GoalVMC/DynBuffEncoder/conv2/kernel'])
  torch.testing.assert_close(grads3['GoalVMC/ConvEncoder/conv2/kernel'], grads['GoalVMC/ConvEncoder/conv2/kernel'], rtol=1e-12, atol=1e-15)


def test_relu_tap_every_branch():
  """``ReluTap`` (mask-consistent gradients of the PLAIN ``loss_and_grads``: the standard of tests/test_model_gpu.py) for
  every proc_obs x proc_tgt branch, the K-step e2e_vmc, velocity heads and L2 > 0: (1) the recording pass names the
  ``conv_encoder`` calls per scope in graph order (target frame first in the sequence branches, graph.py:354); (2) fed the
  restatement's own decisions back, loss and every gradient are reproduced exactly with zero disagreements; (3) one flipped
  decision is counted with its |z| and moves that encoder's gradients; (4) the chunked evaluation's ``plain_grads`` and
  ``conv8`` outputs equal the plain restatement's."""
  cases = [(True, dict(proc_obs='dynimg', proc_tgt='dyndiff', window_size=2), {'ConvEncoder': 1, 'DynBuffEncoder': 1, 'DynDiffEncoder': 1}),
           (True, dict(proc_obs='sequence', proc_tgt='constant', window_size=2), {'ConvEncoder': 3}),
           (True, dict(proc_obs='sequence', proc_tgt='residual', window_size=2, l2_regularizer=1e-3), {'ConvEncoder': 3}),
           (True, dict(proc_obs='sequence', proc_tgt='dyndiff', window_size=2), {'ConvEncoder': 2, 'DynDiffEncoder': 2}),
           (False, dict(window_size=3, control_mode='velocity'), {'ConvEncoder': 3})]
  for goal, kw, calls in cases:
    cfg = O.make_config(img_height=136, img_width=136, batch_size=2, **kw)
    P = O.init_params(O.model_param_shapes(cfg, goal), seed=3)
    feats, labels = O.synthetic_batch(cfg, goal, 2, seed=5, H=136, W=136)
    tr = O.OracleTrainer(cfg, goal, P, dtype=torch.float64)
    loss, _, grads, _, _ = tr.loss_and_grads(feats, labels)
    rec = O.ReluTap(None)
    loss_r, _, grads_r, _, _ = tr.loss_and_grads(feats, labels, tap=rec)
    assert float(loss_r) == float(loss)
    assert {k.split('/')[-1]: v for k, v in rec.calls.items()} == calls, (kw, rec.calls)
    assert all(len(v) == 8 and v[0].dtype == torch.bool for v in rec.recorded.values())
    tap = O.ReluTap(lambda scope, call: rec.recorded[(scope, call)])
    loss2, _, grads2, _, _ = tr.loss_and_grads(feats, labels, tap=tap)
    assert float(loss2) == float(loss)
    for k in grads:
      torch.testing.assert_close(grads2[k], grads[k], rtol=1e-12, atol=1e-15, msg=k)
    for scope, st in tap.stats.items():
      assert all(n == 0 and z == 0.0 and tot > 0 for n, z, tot in st), (scope, st)
    # flip one decision of conv2 in the LAST call of the first encoder
    sc = next(iter(rec.calls))
    m = rec.recorded[(sc, rec.calls[sc] - 1)][1]
    m[(1, 5, 7, 3)] = ~m[(1, 5, 7, 3)]
    tap3 = O.ReluTap(lambda scope, call: rec.recorded[(scope, call)])
    _, _, grads3, _, _ = tr.loss_and_grads(feats, labels, tap=tap3)
    # (forced forward: the flip is counted at its layer; layers above it see a changed input and may disagree too)
    assert tap3.stats[sc][0][0] == 0 and tap3.stats[sc][1][0] == 1 and tap3.stats[sc][1][1] > 1e-6
    assert not torch.equal(grads3[sc + '/conv2/kernel'], grads[sc + '/conv2/kernel'])
  # (4)
  cfg = O.make_config(img_height=136, img_width=136, batch_size=2, proc_obs='dynimg', proc_tgt='dyndiff', window_size=2)
  P = O.init_params(O.model_param_shapes(cfg, True), seed=3)
  feats, labels = O.synthetic_batch(cfg, True, 2, seed=5, H=136, W=136)
  tr = O.OracleTrainer(cfg, True, P, dtype=torch.float64)
  rec = O.ReluTap(None)
  _, _, grads, _, _ = tr.loss_and_grads(feats, labels, tap=rec)
  lines = []
  _, _, g2, _, ep2 = O.loss_and_grads_chunked(tr, feats, labels, chunk=1, plain_grads=True, progress=lines.append,
                                              masks_fn=lambda j, i0, i1: [m[i0:i1] for m in rec.recorded[(list(rec.calls)[j], 0)]])
  assert len(lines) == 12 and lines[0].startswith('forward') and lines[-1].startswith('backward')
  for k, g in ep2['plain_grads'].items():
    torch.testing.assert_close(g, grads[k], rtol=1e-9, atol=1e-14, msg=k)
    torch.testing.assert_close(g2[k], grads[k], rtol=1e-9, atol=1e-14, msg=k)
  assert sorted(ep2['conv8']) == sorted(rec.calls) and all(v.shape == (2, 2, 2, 256) for v in ep2['conv8'].values())


# ----------------------------------------------------------------------------------------------------------------------
# The one vector in this tree that TensorFlow itself wrote: TF 1.15's own unit test
# tensorflow/python/kernel_tests/rnn_cell_test.py::testBasicLSTMCell publishes, to 1e-2 (its assertAllClose tolerance there),
# the result of two stacked cells of 2 units (state_is_tuple=False), every kernel entry 0.5, zero biases, x = [[1, 1]],
# state = 0.1 * ones(1, 8):  output [[0.24024698, 0.24024698]] and new state
# [[0.68967271, 0.68967271, 0.44848421, 0.44848421, 0.39897051, 0.39897051, 0.24024698, 0.24024698]] (layout c1 | h1 | c2 | h2).
# What it pins: the cell formula c' = sigmoid(f + forget_bias) c + sigmoid(i) tanh(j), h' = sigmoid(o) tanh(c') with
# forget_bias = 1 (0 would give c1 = 0.674...), a NON-zero incoming state, and the c-before-h layout of the non-tuple state.
# What it does NOT pin: the gate order i, j, f, o -- all four pre-activations are equal when every kernel entry is 0.5.
# (LSTMCell, which the reference uses at graph.py:217, and BasicLSTMCell share this formula when peepholes, clipping and
# projection are off, as they are there.)  tests/test_kernels_gpu.py runs the same vector through the HIP entry points.
# ----------------------------------------------------------------------------------------------------------------------
TF_BASIC_LSTM_OUTPUT = [[0.24024698, 0.24024698]]
TF_BASIC_LSTM_STATE = [[0.68967271, 0.68967271, 0.44848421, 0.44848421, 0.39897051, 0.39897051, 0.24024698, 0.24024698]]
TF_BASIC_LSTM_TOL = 1e-2        # what the TF test itself asserts; the digits above are its literals


def tf_basic_lstm_case(dtype=torch.float64):
  x = torch.ones(1, 2, dtype=dtype)
  state = torch.full((1, 8), 0.1, dtype=dtype)
  kernel = torch.full((4, 8), 0.5, dtype=dtype)      # [x | h] (2 + 2) x 4 gates of 2 units
  bias = torch.zeros(8, dtype=dtype)
  return x, state, kernel, bias


def test_lstm_cell_reproduces_the_vector_tf_publishes():
  x, state, kernel, bias = tf_basic_lstm_case()
  c1, h1 = O.lstm_cell(x, state[:, 0:2], state[:, 2:4], kernel, bias)
  c2, h2 = O.lstm_cell(h1, state[:, 4:6], state[:, 6:8], kernel, bias)
  new_state = torch.cat([c1, h1, c2, h2], 1)
  np.testing.assert_allclose(h2.numpy(), TF_BASIC_LSTM_OUTPUT, atol=TF_BASIC_LSTM_TOL, rtol=0)
  np.testing.assert_allclose(new_state.numpy(), TF_BASIC_LSTM_STATE, atol=TF_BASIC_LSTM_TOL, rtol=0)
  # the literals carry eight digits of TF's float32 result: the fp64 restatement agrees with them to float32 rounding
  np.testing.assert_allclose(new_state.numpy(), TF_BASIC_LSTM_STATE, atol=2e-7, rtol=0)
  # forget_bias = 0 is NOT the published vector
  c1_0, _ = O.lstm_cell(x, state[:, 0:2], state[:, 2:4], kernel, bias, forget_bias=0.0)
  assert abs(float(c1_0[0, 0]) - TF_BASIC_LSTM_STATE[0][0]) > TF_BASIC_LSTM_TOL


# ----------------------------------------------------------------------------------------------------------------------
# More vectors TensorFlow itself wrote (literals of its own unit tests, TF 1.15; inputs there are "running numbers from 1" in
# NHWC / HWIO order).  They pin exactly the conventions SURVEY 8c lists as "from source knowledge":
#   * tensorflow/python/kernel_tests/conv_ops_test.py: testConv2D2x2FilterStride2Same, testConv2DKernelSmallerThanStrideSame
#     (three cases), testConv2D1x1Filter -> padding='SAME' puts the odd pad element AFTER (bottom / right), out = ceil(in / s);
#   * tensorflow/python/kernel_tests/losses_test.py: MeanSquaredErrorTest.testNonZeroLoss (49.5) -> mean over ALL elements;
#     SoftmaxCrossEntropyLossTest.testAllWrong (10.0 to 3 places) / testAllCorrect (0) -> mean over the batch, one-hot labels.
# The oracle's conv2d_same / same_pad take any kernel size, so TF's 1 x 1 and 2 x 2 cases go through it as published; the HIP
# path (3 x 3 only) runs the stride-2 and stride-3 cases with the filter embedded in a 3 x 3 kernel (tests/test_kernels_gpu.py).
# ----------------------------------------------------------------------------------------------------------------------
TF_CONV_SAME_CASES = [      # (input sizes NHWC, filter sizes HWIO, stride, expected output flattened NHWC)
    ([1, 2, 3, 3], [2, 2, 3, 3], 2, [2271.0, 2367.0, 2463.0, 1230.0, 1305.0, 1380.0]),      # testConv2D2x2FilterStride2Same
    ([1, 3, 3, 1], [1, 1, 1, 1], 2, [1.0, 3.0, 7.0, 9.0]),                                    # testConv2DKernelSmallerThanStrideSame
    ([1, 4, 4, 1], [1, 1, 1, 1], 2, [1.0, 3.0, 9.0, 11.0]),
    ([1, 4, 4, 1], [2, 2, 1, 1], 3, [44.0, 28.0, 41.0, 16.0]),
    ([1, 2, 3, 3], [1, 1, 3, 3], 1, [30.0, 36.0, 42.0, 66.0, 81.0, 96.0, 102.0, 126.0, 150.0, 138.0, 171.0, 204.0, 174.0, 216.0,
                                      258.0, 210.0, 261.0, 312.0]),                          # testConv2D1x1Filter (SAME == VALID for 1 x 1)
]
TF_MSE_PREDICTIONS = [[4.0, 8.0, 12.0], [8.0, 1.0, 3.0]]       # MeanSquaredErrorTest
TF_MSE_LABELS = [[1.0, 9.0, 2.0], [-5.0, -2.0, 6.0]]
TF_MSE_LOSS = 49.5
TF_XENT_LOGITS = [[10.0, 0.0, 0.0], [0.0, 10.0, 0.0], [0.0, 0.0, 10.0]]      # SoftmaxCrossEntropyLossTest
TF_XENT_WRONG_CLASSES = [2, 0, 1]       # one-hot rows [0,0,1], [1,0,0], [0,1,0]
TF_XENT_WRONG_LOSS = 10.0               # assertAlmostEqual(..., 10.0, 3): exact value ln(e^10 + 2) = 10.0000908


def tf_running_numbers(sizes, dtype=torch.float64):
  n = int(np.prod(sizes))
  return torch.arange(1, n + 1, dtype=dtype).reshape(sizes)


def test_conv_same_reproduces_the_vectors_tf_publishes():
  for in_sizes, f_sizes, stride, expected in TF_CONV_SAME_CASES:
    x, w = tf_running_numbers(in_sizes), tf_running_numbers(f_sizes)
    y = O.conv2d_same(x, w, torch.zeros(f_sizes[3], dtype=torch.float64), stride, relu=False)
    assert list(y.shape[1:3]) == [-(-in_sizes[1] // stride), -(-in_sizes[2] // stride)]
    np.testing.assert_array_equal(y.reshape(-1).numpy(), expected)
  # the convention the stride-2 layers of the encoder depend on, said out loud: even input, 3 x 3, stride 2 -> (0 before, 1 after);
  # TF's 2 x 2 / stride 3 case on 4 pixels has the same (0, 1) split, its 2 x 2 / stride 2 case on 3 columns too
  assert O.same_pad(4, 2, 3) == (2, 0, 1) and O.same_pad(3, 2, 2) == (2, 0, 1) and O.same_pad(256, 3, 2) == (128, 0, 1)
  # the PyTorch convention (pad 1 before) gives a different answer on TF's stride-3 case
  x, w = tf_running_numbers([1, 4, 4, 1]), tf_running_numbers([2, 2, 1, 1])
  wrong = torch.nn.functional.conv2d(torch.nn.functional.pad(x.permute(0, 3, 1, 2), (1, 0, 1, 0)), w.permute(3, 2, 0, 1), stride=3)
  assert wrong.reshape(-1).tolist() != [44.0, 28.0, 41.0, 16.0]


def test_losses_reproduce_the_values_tf_publishes():
  pred, lab = torch.tensor(TF_MSE_PREDICTIONS, dtype=torch.float64), torch.tensor(TF_MSE_LABELS, dtype=torch.float64)
  assert float(O.mse(pred, lab)) == TF_MSE_LOSS                      # 297 / 6: a mean over the batch alone would give 148.5
  logits = torch.tensor(TF_XENT_LOGITS, dtype=torch.float64)
  wrong = float(O.softmax_xent(logits, torch.tensor(TF_XENT_WRONG_CLASSES), 3))
  assert abs(wrong - TF_XENT_WRONG_LOSS) < 5e-4 and abs(wrong - math.log(math.exp(10.0) + 2.0)) < 1e-12      # a sum over the batch: 30
  right = float(O.softmax_xent(10.0 * logits, torch.tensor([0, 1, 2]), 3))      # testAllCorrect: logits +-100 -> 0.0 to 3 places
  assert abs(right) < 5e-4


# ----------------------------------------------------------------------------------------------------------------------
# tf.train.AdamOptimizer: TF 1.15's own unit test (tensorflow/python/training/adam_test.py, testBasic) runs var0 = [1, 2],
# var1 = [3, 4] with the constant gradients [0.1, 0.1] / [0.01, 0.01] for three steps at the default lr = 0.001 and compares
# every step against a numpy reference it carries (adam_update_numpy: alpha_t = alpha sqrt(1 - beta2^t) / (1 - beta1^t);
# param - alpha_t m_t / (sqrt(v_t) + epsilon)).  It publishes no literals, so this pins the PROTOCOL and the formula TF
# holds itself to (epsilon beside sqrt(v), "epsilon hat" of its docstring), not numbers; the literal values below follow from
# that formula in closed form for a constant gradient: m_t / (1 - beta1^t) = g and v_t / (1 - beta2^t) = g^2, so every step moves
# the parameter by lr g / (|g| + eps / sqrt(1 - beta2^t)) -- 3 x 0.001 up to the epsilon term.
# ----------------------------------------------------------------------------------------------------------------------
TF_ADAM_VARS = [[1.0, 2.0], [3.0, 4.0]]
TF_ADAM_GRADS = [[0.1, 0.1], [0.01, 0.01]]


def tf_adam_update_numpy(param, g_t, t, m, v, alpha=0.001, beta1=0.9, beta2=0.999, epsilon=1e-8):
  """The reference TF's adam_test.py compares against, restated."""
  alpha_t = alpha * np.sqrt(1 - beta2 ** t) / (1 - beta1 ** t)
  m_t = beta1 * m + (1 - beta1) * g_t
  v_t = beta2 * v + (1 - beta2) * g_t * g_t
  return param - alpha_t * m_t / (np.sqrt(v_t) + epsilon), m_t, v_t


def test_adam_follows_the_protocol_of_tfs_own_test():
  for var0, g0 in zip(TF_ADAM_VARS, TF_ADAM_GRADS):
    ref, rm, rv = np.array(var0), np.zeros(2), np.zeros(2)
    p, m, v = np.array(var0), np.zeros(2), np.zeros(2)
    g = np.array(g0)
    for t in (1, 2, 3):
      ref, rm, rv = tf_adam_update_numpy(ref, g, t, rm, rv)
      O.adam_step_tf(p, g, m, v, t, 0.001)
      np.testing.assert_allclose(p, ref, rtol=1e-14, atol=0)
      np.testing.assert_allclose(m, rm, rtol=1e-14); np.testing.assert_allclose(v, rv, rtol=1e-14)
      closed = np.array(var0) - sum(0.001 * g / (np.abs(g) + 1e-8 / math.sqrt(1 - 0.999 ** k)) for k in range(1, t + 1))
      np.testing.assert_allclose(p, closed, rtol=1e-12)
    assert np.all(np.abs((np.array(var0) - p) - 0.003) < 1e-7)
