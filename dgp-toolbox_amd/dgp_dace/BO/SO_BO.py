"""Single-objective (constrained) Bayesian optimisation on this package's surrogates — host-side mirror of the
reference's `dgp_dace/BO/SO_BO.py` (the caller of the DGP hot path): same class, constructor arguments, attributes
(`X, Y, C, X_n, lw_n, up_n, X_train, Ymin, Xfeasible, model_Y, model_C, IC, added_points, ...`) and methods
(`feasible, make_model, train_model, train_models, run, add_point`).  What differs:

* surrogates are `dgp_dace.models.dgp.DGP` (num_layers > 0) and `dgp_dace.models.gpr.GPR` (num_layers == 0) on the
  MI355X engine; the exact GP is trained by `GPR.optimize_adam` where the reference drives `tf.optimizers.Adam()`;
* the initial design is `scipy.stats.qmc.LatinHypercube` (the reference calls `pyDOE.lhs`, not installed here);
* three slips of the reference's `run` are not reproduced: a re-created objective model is assigned (SO_BO.py:280 drops
  it), every constraint model is re-fed its own column (:291 reshapes all columns into one), and `constraint_handling`
  'PoF' is refused (Infill_criteria.PoF.run returns nothing there).
"""
import numpy as np
from scipy.stats import qmc

from ..Infill_criteria import EI, EV, WB2, WB2S
from ..gpflow_compat import Gaussian, Matern32, Matern52, SquaredExponential
from ..models.dgp import DGP
from ..models.gpr import GPR

_KERNELS = {'rbf': SquaredExponential, 'matern32': Matern32, 'matern52': Matern52}


def normalize(*args):
    out = [(a - a.mean(axis=0)) / a.std(axis=0) for a in args]
    return out[0] if len(out) == 1 else out


def normalize_X(X):
    m, s = X.mean(axis=0), X.std(axis=0)
    return (X - m) / s, (0 - m) / s, (1 - m) / s


def normalize_C(C):
    m, s = C.mean(axis=0), C.std(axis=0)
    return (C - m) / s, (0 - m) / s


def denormalize(Xstar_N, X):
    return X.std(axis=0) * Xstar_N + X.mean(axis=0)


def denormalize_var(Xstar_N, X):
    return X.std(axis=0) ** 2 * Xstar_N


def DoE(problem, DoE_size, seed=None):
    X = qmc.LatinHypercube(d=problem.dim, seed=seed).random(DoE_size)
    if problem.constraint:
        Y, C = problem.fun(X)
        return X, Y, C
    return X, problem.fun(X)[0]


def _kernel(name, dim):
    if name not in _KERNELS:
        raise Exception("The kernel has to be a string or a list of strings: 'rbf', 'matern32', matern52'")
    return _KERNELS[name](lengthscales=[1.0] * dim, variance=1.0)


class SO_BO(object):
    def __init__(self, problem=None, X=None, Y=None, C=None, DoE_size=None, model_Y_dic=None, model_C_dic=None,
                 normalize_input=True, seed=None):
        if problem is None:
            raise Exception("You have to specify a problem to optimize")
        if not isinstance(model_Y_dic, dict):
            raise Exception("You have to specify a dictionary for the architecture of the objective function model")
        if problem.constraint and model_C_dic is None:
            raise Exception("You have to specify a dictionary for the architecture of the constraint functions models")
        if DoE_size is None and X is None:
            raise Exception("You have to specify either a size to generate a DoE or specify a known DoE (X,Y)")
        self.problem, self.model_Y_dic, self.model_C_dic = problem, model_Y_dic, model_C_dic
        self.seed = seed
        if X is None:
            data = DoE(problem, DoE_size, seed)
            self.X, self.Y = data[0], data[1]
            self.C = data[2] if problem.constraint else None
        else:
            self.X, self.Y = np.array(X, dtype=np.float64), np.array(Y, dtype=np.float64)
            self.C = np.array(C, dtype=np.float64) if problem.constraint else None
        self.d, self.n = problem.dim, self.X.shape[0]
        self.normalize_input = normalize_input
        self._refresh_training_arrays()
        self.model_Y = self.make_model(model_Y_dic, self.X_train, self.Y_train)
        if problem.constraint:
            if not isinstance(model_C_dic, list):
                self.model_C_dic = [model_C_dic] * self.C.shape[1]
            for dic in self.model_C_dic:
                if not isinstance(dic, dict):
                    raise Exception("every entry of model_C_dic has to be a dictionary")
            self.model_C = [self.make_model(self.model_C_dic[i], self.X_train, self.C_train[:, i:i + 1])
                            for i in range(self.C.shape[1])]
        self.Xfeasible, self.Yfeasible, self.Ymin = [], [], []
        self.feasible()
        self.added_points = []
        self.IC = None
        self.constrained_IC = None

    # ------------------------------------------------------------------ data
    def _refresh_training_arrays(self):
        if self.normalize_input:
            self.X_n, self.lw_n, self.up_n = normalize_X(self.X)
            self.Y_n = normalize(self.Y)
            if self.C is not None:
                self.C_n, self.feasible_0 = normalize_C(self.C)
            self.X_train, self.Y_train = self.X_n, self.Y_n
            self.C_train = self.C_n if self.C is not None else None
        else:
            self.lw_n, self.up_n = np.zeros(self.d), np.ones(self.d)
            self.X_train, self.Y_train, self.C_train = self.X, self.Y, self.C
            if self.C is not None:
                self.feasible_0 = np.zeros(self.C.shape[1])

    def feasible(self):
        """Feasible observations (every constraint <= 0) and the best feasible value so far (SO_BO.py:155-176)."""
        if self.C is None:
            self.Xfeasible, self.Yfeasible, self.Ymin = self.X, self.Y, [np.min(self.Y)]
            return
        ok = self.C.max(axis=1) <= 0
        self.Xfeasible, self.Yfeasible, self.Cfeasible = self.X[ok].ravel(), self.Y[ok].ravel(), self.C[ok].ravel()
        self.Ymin = [np.min(self.Yfeasible)] if ok.any() else [np.max(self.Y)]

    # ------------------------------------------------------------------ models
    def make_model(self, dic, X, Y):
        """A surrogate from an architecture dictionary {'num_layers', 'num_units', 'kernels', 'num_samples'}
        (SO_BO.py:177-250): exact GP for num_layers == 0, otherwise a DGP whose inducing inputs are the data."""
        num_layers = dic['num_layers']
        kern_names = dic['kernels']
        if num_layers == 0:
            return GPR((X, Y), _kernel(kern_names, X.shape[1]), noise_variance=1e-5)
        num_units = dic['num_units']
        if isinstance(num_units, int):
            num_units = [num_units] * num_layers
        elif not (isinstance(num_units, list) and len(num_units) == num_layers):
            raise Exception("num_units has to be an integer or a list with one integer per layer")
        if isinstance(kern_names, str):
            kern_names = [kern_names] * (num_layers + 1)
        elif not (isinstance(kern_names, list) and len(kern_names) == num_layers + 1):
            raise Exception("kernels has to be a string or a list with one string per layer plus one")
        dims = [X.shape[1]] + list(num_units)
        kernels = [_kernel(kern_names[l], dims[l]) for l in range(num_layers + 1)]
        return DGP(X, Y, X, kernels, num_units, Gaussian(), num_samples=dic['num_samples'])

    def train_model(self, model, iteration=3000):
        if model.name == 'gpr':
            model.optimize_adam(iterations=iteration)                     # tf.optimizers.Adam() defaults (SO_BO.py:252-256)
        if model.name == 'dgp':
            model.optimize_nat_adam(iterations1=500, iterations2=iteration, beta_1=0.8, beta_2=0.9, lr_gamma=0.01)

    def train_models(self, iteration_Y=3000, iteration_C=3000):
        print('Training of the objective function model')
        self.train_model(self.model_Y, iteration_Y)
        if self.problem.constraint:
            if not isinstance(iteration_C, list):
                iteration_C = [iteration_C] * self.C.shape[1]
            for i in range(self.C.shape[1]):
                print('Training of constraint model', i + 1)
                self.train_model(self.model_C[i], iteration_C[i])

    # ------------------------------------------------------------------ the loop (SO_BO.py:270-313)
    def run(self, iterations, from_scratch=None, IC='EI', constraint_handling='EV', threshold=0.1, train_iterations=1000,
            popsize_DE=300, popstd_DE=1.5, iterations_DE=400, init_adam=None, iterations_adam=1000, IC_method='DE+Adam',
            analytic=True):
        if constraint_handling == 'PoF' and self.problem.constraint:
            raise NotImplementedError("constraint_handling='PoF': the reference's PoF criterion is unfinished; use 'EV'")
        criteria = {'EI': EI, 'WB2': WB2, 'WB2S': WB2S}
        if from_scratch is None:
            from_scratch = iterations + 1
        for j in range(iterations):
            print('adding the most promising data point in iteration', j)
            bounds = (self.lw_n, self.up_n)
            if j % from_scratch == 0 and j != 0:
                self.model_Y = self.make_model(self.model_Y_dic, self.X_train, self.Y_train)
                if self.problem.constraint:
                    self.model_C = [self.make_model(self.model_C_dic[i], self.X_train, self.C_train[:, i:i + 1])
                                    for i in range(self.C.shape[1])]
            if j % from_scratch == 0:
                self.train_models(iteration_Y=train_iterations, iteration_C=train_iterations)
            elif j != 0:
                self.model_Y.data = (self.X_train, self.Y_train)
                if self.problem.constraint:
                    for i in range(self.C.shape[1]):
                        self.model_C[i].data = (self.X_train, self.C_train[:, i:i + 1])
                self.train_models(iteration_Y=int(train_iterations / 2), iteration_C=int(train_iterations / 2))
            y_min_n = (self.Ymin[-1] - self.Y.mean(axis=0)) / self.Y.std(axis=0)
            self.IC = criteria[IC](y_min_n, self.d)
            if self.problem.constraint:
                self.constrained_IC = EV(self.feasible_0, self.d)
                self.added_points = self.constrained_IC.optimize_with_IC(
                    self.IC, self.model_Y, self.model_C, bounds, threshold=threshold, popsize_DE=popsize_DE,
                    popstd_DE=popstd_DE, iterations_DE=iterations_DE, iterations_adam=iterations_adam, method=IC_method,
                    analytic=analytic)
            else:
                kw = dict(analytic=analytic) if IC == 'EI' else {}
                self.added_points = self.IC.optimize(self.model_Y, bounds, popsize_DE=popsize_DE, popstd_DE=popstd_DE,
                                                     iterations_DE=iterations_DE, init_adam=init_adam,
                                                     iterations_adam=iterations_adam, method=IC_method, **kw)
            self.add_point()
            print('Actual Y min:', self.Ymin[-1])

    def add_point(self):
        """Evaluate the problem at the proposed point, append, refresh normalisation and feasibility (SO_BO.py:315-350)."""
        x_new = np.asarray(self.added_points, dtype=np.float64).reshape(1, self.d)
        if self.normalize_input:
            x_new = denormalize(x_new, self.X)
        res = self.problem.fun(x_new)
        self.X = np.append(self.X, x_new, axis=0)
        self.Y = np.append(self.Y, np.asarray(res[0]).reshape(1, -1), axis=0)
        if self.problem.constraint:
            self.C = np.append(self.C, np.asarray(res[1]).reshape(1, -1), axis=0)
            if self.C[-1].max() <= 0:
                self.Yfeasible = np.append(self.Yfeasible, self.Y[-1])
                self.Xfeasible = np.append(self.Xfeasible, self.X[-1])
                self.Ymin = np.append(self.Ymin, np.min(self.Yfeasible))
            else:
                self.Ymin = np.append(self.Ymin, self.Ymin[-1])
        else:
            self.Yfeasible, self.Xfeasible = self.Y, self.X
            self.Ymin = np.append(self.Ymin, np.min(self.Y))
        self._refresh_training_arrays()
