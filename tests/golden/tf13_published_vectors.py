"""Known-answer vectors published in TensorFlow's own unit tests (the un-vendored dependency the reference pins:
tensorflow==1.3.0, requirements-cpu.txt:1).  Transcribed offline from
  tensorflow/python/kernel_tests/conv_ops_test.py     (Conv2DTest: testConv2D*, *BackpropInput, *BackpropFilter)
  tensorflow/python/kernel_tests/pooling_ops_test.py  (_testMaxPoolValidPadding)
  tensorflow/python/ops/image_ops_test.py             (ResizeImagesTest.testResizeUp / testResizeDown, BILINEAR)
  tensorflow/python/kernel_tests/extract_image_patches_op_test.py (testKsize2x2Stride1x1Rate1x1Valid / ...Same)
In those tests every operand is filled with 1, 2, 3, ... in row-major order (`x = [f * 1.0 for f in range(1, n + 1)]`)
unless data is given.  Round 2 adds
  tensorflow/python/training/adam_test.py                  (AdamOptimizerTest.testBasic: inputs + its numpy recurrence)
  tensorflow/python/kernel_tests/pooling_ops_test.py       (_testMaxPoolGradDirect1: ties go to the first maximum)
  tensorflow/python/kernel_tests/histogram_ops_test.py / histogram_fixed_width's docstring example
  tensorflow/python/ops/nn_test.py                         (DropoutTest: kept elements are x / keep_prob)
Round 3 adds
  tensorflow/python/training/gradient_descent_test.py      (GradientDescentOptimizerTest.testBasic)
  tf.scatter_nd_update's documented example (python/ops/state_ops.py docstring): rank-1 update of 4 of 8 elements
These are DATA (inputs by rule + expected outputs), not source; they pin the oracle's conv2d
(VALID / SAME incl. the odd SAME split, stride > kernel), both conv gradients, max-pool and the legacy bilinear resize
to what TensorFlow 1.3 computes.
"""

# (input NHWC shape, filter HWIO shape, stride, padding, expected flat output)
CONV2D_FWD = [
    ('testConv2D1x1Filter', (1, 2, 3, 3), (1, 1, 3, 3), 1, 'VALID',
     [30.0, 36.0, 42.0, 66.0, 81.0, 96.0, 102.0, 126.0, 150.0, 138.0, 171.0, 204.0, 174.0, 216.0, 258.0, 210.0, 261.0,
      312.0]),
    ('testConv2D2x2Filter', (1, 2, 3, 3), (2, 2, 3, 3), 1, 'VALID', [2271.0, 2367.0, 2463.0, 2901.0, 3033.0, 3165.0]),
    ('testConv2D2x2FilterStride2', (1, 2, 3, 3), (2, 2, 3, 3), 2, 'VALID', [2271.0, 2367.0, 2463.0]),
    ('testConv2D2x2FilterStride2Same', (1, 2, 3, 3), (2, 2, 3, 3), 2, 'SAME',
     [2271.0, 2367.0, 2463.0, 1230.0, 1305.0, 1380.0]),
    ('testConv2DKernelSizeMatchesInputSize', (1, 2, 2, 1), (2, 2, 1, 2), 1, 'VALID', [50.0, 60.0]),
    ('testConv2DKernelSmallerThanStrideSame/3x3', (1, 3, 3, 1), (1, 1, 1, 1), 2, 'SAME', [1.0, 3.0, 7.0, 9.0]),
    ('testConv2DKernelSmallerThanStrideSame/4x4', (1, 4, 4, 1), (1, 1, 1, 1), 2, 'SAME', [1.0, 3.0, 9.0, 11.0]),
    ('testConv2DKernelSmallerThanStrideSame/2x2s3', (1, 4, 4, 1), (2, 2, 1, 1), 3, 'SAME', [44.0, 28.0, 41.0, 16.0]),
]

# (input shape, filter shape (values 1..), out_backprop shape (values 1..), stride, padding, expected d input)
CONV2D_BACKPROP_INPUT = [
    ('testConv2D2x2Depth1ValidBackpropInput', (1, 2, 3, 1), (2, 2, 1, 1), (1, 1, 2, 1), 1, 'VALID',
     [1.0, 4.0, 4.0, 3.0, 10.0, 8.0]),
    ('testConv2D2x2Depth3ValidBackpropInput', (1, 2, 3, 3), (2, 2, 3, 3), (1, 1, 2, 3), 1, 'VALID',
     [14.0, 32.0, 50.0, 100.0, 163.0, 226.0, 167.0, 212.0, 257.0, 122.0, 140.0, 158.0, 478.0, 541.0, 604.0, 437.0,
      482.0, 527.0]),
]

# (input shape (values 1..), filter shape, out_backprop shape (values 1..), stride, padding, expected d filter)
CONV2D_BACKPROP_FILTER = [
    ('testConv2D2x2Depth1ValidBackpropFilter', (1, 2, 3, 1), (2, 2, 1, 1), (1, 1, 2, 1), 1, 'VALID',
     [5.0, 8.0, 14.0, 17.0]),
    ('testConv2D2x2Depth3ValidBackpropFilter', (1, 2, 3, 3), (2, 2, 3, 3), (1, 1, 2, 3), 1, 'VALID',
     [17.0, 22.0, 27.0, 22.0, 29.0, 36.0, 27.0, 36.0, 45.0, 32.0, 43.0, 54.0, 37.0, 50.0, 63.0, 42.0, 57.0, 72.0, 62.0,
      85.0, 108.0, 67.0, 92.0, 117.0, 72.0, 99.0, 126.0, 77.0, 106.0, 135.0, 82.0, 113.0, 144.0, 87.0, 120.0, 153.0]),
]

# 2x2 / stride 2 VALID max pool of a (1,3,3,3) tensor 1..27
MAXPOOL_VALID = ((1, 3, 3, 3), [13.0, 14.0, 15.0])

# tf.image.resize_images(..., BILINEAR) (align_corners=False): (input shape, data, target h, w, expected)
RESIZE_BILINEAR = [
    ('testResizeUp', (1, 3, 2, 1), [64, 32, 32, 64, 50, 100], 6, 4,
     [64.0, 48.0, 32.0, 32.0, 48.0, 48.0, 48.0, 48.0, 32.0, 48.0, 64.0, 64.0, 41.0, 61.5, 82.0, 82.0, 50.0, 75.0,
      100.0, 100.0, 50.0, 75.0, 100.0, 100.0]),
    ('testResizeDown', (1, 6, 4, 1),
     [128, 128, 64, 64, 128, 128, 64, 64, 64, 64, 128, 128, 64, 64, 128, 128, 50, 50, 100, 100, 50, 50, 100, 100], 3, 2,
     [128.0, 64.0, 64.0, 128.0, 50.0, 100.0]),
]

# tf.extract_image_patches of image [[1, 2], [3, 4]] (1,2,2,1), ksize 2x2, stride 1: (padding, expected [1,rows,cols,4])
EXTRACT_PATCHES_2X2 = [
    ('VALID', [[[[1, 2, 3, 4]]]]),
    ('SAME', [[[[1, 2, 3, 4], [2, 0, 4, 0]], [[3, 4, 0, 0], [4, 0, 0, 0]]]]),
]


# AdamOptimizerTest.testBasic: two variables, constant gradients, default hyper-parameters, three steps.  The test's
# expected values are its own numpy recurrence (adam_update_numpy), reproduced by the consuming tests from these inputs:
#   alpha_t = lr * sqrt(1 - beta2**t) / (1 - beta1**t);  m_t = beta1*m + (1-beta1)*g;  v_t = beta2*v + (1-beta2)*g*g
#   param_t = param - alpha_t * m_t / (sqrt(v_t) + epsilon);   assertAllCloseAccordingToType (float32: 1e-6)
ADAM_TEST_BASIC = {
    'var0': [1.0, 2.0], 'grads0': [0.1, 0.1], 'var1': [3.0, 4.0], 'grads1': [0.01, 0.01],
    'lr': 0.001, 'beta1': 0.9, 'beta2': 0.999, 'epsilon': 1e-8, 'steps': 3,
}

# _testMaxPoolGradDirect1 ("constant gradient behavior"): input 1x4x4x1 of ones, 2x2 window, stride 1, VALID; every window
# is a four-way tie and MaxPoolGrad sends the window's gradient to its FIRST element in scan order.
MAXPOOL_GRAD_DIRECT1 = {
    'input_sizes': (1, 4, 4, 1), 'input_data': [1.0] * 16, 'window': 2, 'stride': 1,
    'output_backprop': [11.0, 12.0, 13.0, 15.0, 16.0, 17.0, 19.0, 20.0, 21.0],
    'expected_input_backprop': [11.0, 12.0, 13.0, 0.0, 15.0, 16.0, 17.0, 0.0, 19.0, 20.0, 21.0, 0.0, 0.0, 0.0, 0.0, 0.0],
}

# tf.histogram_fixed_width: values below the range fall into the first bin, values >= the upper edge into the last.
HISTOGRAM_FIXED_WIDTH = {
    'value_range': [0.0, 5.0], 'nbins': 5, 'new_values': [-1.0, 0.0, 1.5, 2.0, 5.0, 15.0], 'expected': [2, 1, 1, 0, 2],
}

# nn.dropout: binary = floor(keep_prob + U); y = x / keep_prob * binary -> an all-ones input comes out as {0, 1/keep_prob}
DROPOUT_KEEP_PROBS = [0.1, 0.5, 0.8]

# GradientDescentOptimizerTest.testBasic: var -= learning_rate * grad  (dcnf's tf.train.GradientDescentOptimizer(0.1),
# src/models.py:198)
SGD_TEST_BASIC = {'learning_rate': 3.0, 'var0': [1.0, 2.0], 'var1': [3.0, 4.0], 'grads0': [0.1, 0.1], 'grads1': [0.01, 0.01],
                  'expected0': [1.0 - 3.0 * 0.1, 2.0 - 3.0 * 0.1], 'expected1': [3.0 - 3.0 * 0.01, 4.0 - 3.0 * 0.01]}

# tf.scatter_nd_update docstring: ref = [1..8]; indices [[4], [3], [1], [7]]; updates [9, 10, 11, 12]
# (get_A builds the CRF's R with two such updates along (left, right) and (right, left), src/models.py:138-141)
SCATTER_ND_UPDATE_DOC = {'ref': [1, 2, 3, 4, 5, 6, 7, 8], 'indices': [[4], [3], [1], [7]], 'updates': [9, 10, 11, 12],
                         'expected': [1, 11, 3, 10, 9, 6, 7, 12]}
