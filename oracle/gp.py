"""ORACLE (test infrastructure, never shipped): float64 restatement of the reference's GP predictor at prediction time.

Reference: ``GPRegressionModel`` (src/predictors/gp.py:13-32) = gpytorch ``ExactGP`` with ``ConstantMean`` and
``InducingPointKernel(ScaleKernel(RBFKernel()), inducing_points=train_x[:500])`` + ``GaussianLikelihood``; trained and
evaluated in experiments/01_bn_asia/main.py:315-393 (first 80 % of predictor_dataset = train, rest = test,
``preds = model(test_x)``; ``preds.mean``).  An ExactGP whose kernel is the Nystroem kernel Q = K_xu K_uu^-1 K_ux has the
SGPR / DTC predictive mean
    mean(x*) = c + K_*u (K_uu + s^-2 K_uf K_fu)^-1 K_uf (y - c) s^-2,     k(a, b) = o * exp(-|a - b|^2 / (2 l^2)),
with o = softplus(raw_outputscale), l = softplus(raw_lengthscale), s^2 = softplus(raw_noise) + 1e-4 (gpytorch's default
constraints), c = raw_constant.

PARITY UNPINNED: gpytorch (1.13, requirements.txt) is not installed and the reference holds no predictions to compare
with; details gpytorch adds (jitter on K_uu, Cholesky vs CG solves) are not restated.  What the tests check instead: the
device kernel against this file, and that the SHIPPED hyper-parameters reproduce the behaviour the survey measured —
on the reference's own split the test MAE equals the constant predictor's (686.1 vs 686.3), i.e. the shipped GP carries
no information beyond the mean (lengthscale 8.7 and outputscale 19 against targets of -13 500 +- 840).
"""
import numpy as np


def softplus(v):
    v = np.asarray(v, np.float64)
    return np.where(v > 30, v, np.log1p(np.exp(np.minimum(v, 30))))


def hyper(fix):
    return dict(noise=float(softplus(fix["raw_noise"]).reshape(-1)[0]) + 1e-4,
                outputscale=float(softplus(fix["raw_outputscale"]).reshape(-1)[0]),
                lengthscale=float(softplus(fix["raw_lengthscale"]).reshape(-1)[0]),
                constant=float(np.asarray(fix["raw_constant"]).reshape(-1)[0]))


def kernel(a, b, outputscale, lengthscale):
    d2 = ((a[:, None, :] - b[None, :, :]) ** 2).sum(-1)
    return outputscale * np.exp(-0.5 * d2 / lengthscale ** 2)


def fit_alpha(train_x, train_y, Z, h, jitter=1e-6):
    """alpha with mean(x*) = c + K_*u alpha."""
    X, y, Z = (np.asarray(v, np.float64) for v in (train_x, train_y, Z))
    Kuu = kernel(Z, Z, h["outputscale"], h["lengthscale"]) + jitter * np.eye(len(Z))
    Kuf = kernel(Z, X, h["outputscale"], h["lengthscale"])
    A = Kuu + Kuf @ Kuf.T / h["noise"]
    return np.linalg.solve(A, Kuf @ (y - h["constant"])) / h["noise"]


def predict_mean(x, Z, alpha, h):
    return h["constant"] + kernel(np.asarray(x, np.float64), np.asarray(Z, np.float64), h["outputscale"], h["lengthscale"]) @ alpha
