"""Construction of the layer stack (reference: dgp_dace/utils/layer_initializations.py:24-68).

Host-side, once per model, NumPy — as in the reference.  Hidden layers get an Identity mean
function when the width is kept, otherwise a fixed Linear map: the leading PCA directions of the
running inputs when stepping down, identity-plus-zero-padding when stepping up; the running
inducing inputs and data are pushed through the same map.
"""
from __future__ import annotations

import numpy as np

from ..gpflow_compat import Identity, Linear, Zero, kernel_from_any, mean_function_from_any, set_trainable
from .layers import SVGP_Layer


def init_layers_linear(X, Y, Z, kernels, num_units, num_outputs=None, mean_function=None, Layer=SVGP_Layer,
                       white=False):
    num_outputs = num_outputs or Y.shape[1]
    widths = [X.shape[1]] + list(num_units)
    if len(kernels) != len(widths):
        raise Exception("one kernel per layer is required: len(kernels) must equal len(num_units) + 1")
    layers = []
    x_run, z_run = np.array(X, dtype=np.float64), np.array(Z, dtype=np.float64)
    print('The DGP architecture')
    for i, (d_in, d_out) in enumerate(zip(widths[:-1], widths[1:])):
        print('layer', i + 1, ': dim_in', d_in, '--> dim_out', d_out)
        W = None
        if d_in == d_out:
            mf = Identity()
        else:
            if d_in > d_out:
                _, _, Vt = np.linalg.svd(x_run, full_matrices=False)
                W = Vt[:d_out, :].T
            else:
                W = np.concatenate([np.eye(d_in), np.zeros((d_in, d_out - d_in))], 1)
            mf = Linear(W)
            set_trainable(mf, False)
        layers.append(Layer(kernel_from_any(kernels[i], d_in), z_run, d_out, mf, white=white))
        if W is not None:
            z_run, x_run = z_run.dot(W), x_run.dot(W)
    final_mf = mean_function_from_any(mean_function) if mean_function is not None else Zero()
    layers.append(Layer(kernel_from_any(kernels[-1], widths[-1]), z_run, num_outputs, final_mf, white=white))
    return layers
