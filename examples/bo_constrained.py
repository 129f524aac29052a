"""Constrained Bayesian optimisation as in the reference's Notebooks_dgp/nb_dgp_BO.ipynb (cells 4-6, 11, 15, 61):
exact-GP objective model, 2-layer DGP constraint model, expected improvement x expected violation, DE + Adam on the
acquisition.  Everything (training, prediction, the input gradient of the acquisition) runs on the HIP engine.
Run from the repository root:  python examples/bo_constrained.py
"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "dgp-toolbox_amd"))
import numpy as np
from dgp_dace.BO.SO_BO import SO_BO


class Constrained_problem(object):
    def __init__(self):
        self.constraint = True
        self.dim = 1

    def fun(self, x):                       # objective, constraint (feasible where <= threshold)
        return [(x - 0.5) ** 2, np.where(x > 0.25, 1.0, 0.0)]


bo = SO_BO(Constrained_problem(), DoE_size=5, model_Y_dic={'num_layers': 0, 'kernels': 'rbf'},
           model_C_dic={'num_layers': 2, 'num_units': 1, 'kernels': 'rbf', 'num_samples': 10}, seed=1)
bo.run(4, from_scratch=2, IC='EI', train_iterations=1500, popsize_DE=30, popstd_DE=3.0, threshold=0.1, iterations_DE=10,
       constraint_handling='EV', iterations_adam=10, IC_method='DE+Adam', analytic=True)
print("evaluated points:", np.round(bo.X[:, 0], 3))
print("best feasible objective per iteration:", bo.Ymin)
