"""Native model descriptors: the device-side replacement of the `model::GenerativeFunction`
argument of pf_initialize (reference src/initialize.jl:31-35).  A descriptor is just
(model id, parameter vector); the parameter layout is that of csrc/gpf_models.hpp.

The reference ships no benchmark models; the three state-space models of
BASELINE.json and the README's object_motion are defined here, with the synthetic data
generators of SURVEY.md §8d (data seed 20240001)."""
from __future__ import annotations

import math
from dataclasses import dataclass, field

import numpy as np

MODEL_LGSSM2, MODEL_BEARINGS4, MODEL_SV1, MODEL_OBJECT_MOTION, MODEL_LINE = 1, 2, 3, 4, 5
DATA_SEED = 20240001
_HALF_LOG_2PI = 0.5 * math.log(2.0 * math.pi)


@dataclass
class NativeModel:
    model_id: int
    name: str
    dim: int
    obs_dim: int
    params: np.ndarray
    info: dict = field(default_factory=dict)

    def row_width(self, keep_prev: bool) -> int:
        w = 2 * self.dim if keep_prev else self.dim
        return w + (w & 1)


def lgssm2(theta: float = 0.1, rho: float = 0.99, sq: float = 0.1, sr: float = 0.5, s0: float = 1.0) -> NativeModel:
    """x' = A x + N(0, sq^2 I), y = x + N(0, sr^2 I), A = rho * rot(theta), x1 ~ N(0, s0^2 I)."""
    a11, a12 = rho * math.cos(theta), -rho * math.sin(theta)
    a21, a22 = rho * math.sin(theta), rho * math.cos(theta)
    def prop(prior_sd):      # locally optimal proposal given a N(mu, prior_sd^2 I) prior and y = x + N(0, sr^2 I)
        gain = prior_sd ** 2 / (prior_sd ** 2 + sr ** 2)
        sv = math.sqrt(prior_sd ** 2 * sr ** 2 / (prior_sd ** 2 + sr ** 2))
        return [gain, sv, 1.0 / sv, 2.0 * (math.log(sv) + _HALF_LOG_2PI), 1.0 / prior_sd, 2.0 * (math.log(prior_sd) + _HALF_LOG_2PI)]
    p = np.array([a11, a12, a21, a22, sq, s0, 1.0 / sr, 2.0 * (math.log(sr) + _HALF_LOG_2PI)] + prop(sq) + prop(s0))
    return NativeModel(MODEL_LGSSM2, "lgssm2", 2, 2, p,
                       dict(A=np.array([[a11, a12], [a21, a22]]), sq=sq, sr=sr, s0=s0))


def bearings4(mu=(-0.05, 0.2, 0.001, -0.055), s=(0.5, 0.3, 0.005, 0.01), sp: float = 0.001, sv: float = 0.001,
              sb: float = 0.005) -> NativeModel:
    p = np.array(list(mu) + list(s) + [sp, sv, 1.0 / sb, math.log(sb) + _HALF_LOG_2PI])
    return NativeModel(MODEL_BEARINGS4, "bearings4", 4, 1, p, dict(mu=mu, s=s, sp=sp, sv=sv, sb=sb))


def sv1(mu: float = -1.0, phi: float = 0.97, sigma: float = 0.15) -> NativeModel:
    p = np.array([mu, phi, sigma, sigma / math.sqrt(1.0 - phi * phi), _HALF_LOG_2PI])
    return NativeModel(MODEL_SV1, "sv1", 1, 1, p, dict(mu=mu, phi=phi, sigma=sigma))


def object_motion(p_stay: float = 0.75, p_start: float = 0.25, sy: float = 0.01, sobs: float = 0.25) -> NativeModel:
    """README.md:43-55 of the reference.  Per-step data vector = [y_obs, sin(t)]."""
    p = np.array([p_stay, p_start, sy, 1.0 / sobs, math.log(sobs) + _HALF_LOG_2PI,
                  math.log(p_stay), math.log1p(-p_stay), math.log(p_start), math.log1p(-p_start)])   # stratified init/update
    return NativeModel(MODEL_OBJECT_MOTION, "object_motion", 2, 2, p, dict(sy=sy, sobs=sobs, p_stay=p_stay, p_start=p_start,
                                                                          strata_address="moving"))


def line_model(p_out: float = 0.1, s_in: float = 1.0, s_out: float = 10.0, slope_lo: int = -2, slope_hi: int = 2) -> NativeModel:
    """The fixture model of the reference's own tests (test/runtests.jl:3-16): slope ~ uniform_discrete(-2, 2);
    step t: x = t, outlier ~ bernoulli(0.1), y ~ normal(x * slope, outlier ? 10 : 1).  Particle row = (slope, outlier_t);
    per-step data vector = [y_t, x_t] (`line_obs`); x_t = 0 stands for model args (0,): no step yet, nothing observed."""
    n = slope_hi - slope_lo + 1
    p = np.array([p_out, 1.0 / s_in, 1.0 / s_out, math.log(s_in) + _HALF_LOG_2PI, math.log(s_out) + _HALF_LOG_2PI,
                  math.log(p_out), math.log1p(-p_out), -math.log(n), float(slope_lo), float(n)])
    return NativeModel(MODEL_LINE, "line_model", 2, 2, p, dict(p_out=p_out, s_in=s_in, s_out=s_out, slopes=list(range(slope_lo, slope_hi + 1)),
                                                                strata_address={"initialize": "slope", "update": "outlier"}))


def line_obs(t: int, slope: float = 0.0) -> np.ndarray:
    """line_choicemap of the reference's tests for ONE step (test/runtests.jl:22-23): y_t = t * slope, with the step index."""
    return np.array([t * slope, float(t)])


def by_name(name: str) -> NativeModel:
    return {"lgssm2": lgssm2, "bearings4": bearings4, "sv1": sv1, "object_motion": object_motion, "line_model": line_model}[name]()


# ----------------------------------------------------------------------------- synthetic data
def simulate(model: NativeModel, T: int, seed: int = DATA_SEED) -> np.ndarray:
    """One trajectory from the model; returns the (T, obs_dim) array of per-step data vectors."""
    rng = np.random.Generator(np.random.Philox(key=seed))
    P = model.params
    out = np.zeros((T, model.obs_dim))
    if model.model_id == MODEL_LGSSM2:
        A, sq, sr, s0 = model.info["A"], model.info["sq"], model.info["sr"], model.info["s0"]
        x = s0 * rng.standard_normal(2)
        for t in range(T):
            if t > 0:
                x = A @ x + sq * rng.standard_normal(2)
            out[t] = x + sr * rng.standard_normal(2)
    elif model.model_id == MODEL_BEARINGS4:
        x = np.array(P[0:4]) + np.array(P[4:8]) * rng.standard_normal(4)
        for t in range(T):
            if t > 0:
                z = rng.standard_normal(4)
                x = np.array([x[0] + x[2] + P[8] * z[0], x[1] + x[3] + P[8] * z[1], x[2] + P[9] * z[2], x[3] + P[9] * z[3]])
            out[t, 0] = math.atan2(x[1], x[0]) + model.info["sb"] * rng.standard_normal()
    elif model.model_id == MODEL_SV1:
        mu, phi, sigma = P[0], P[1], P[2]
        h = mu + P[3] * rng.standard_normal()
        for t in range(T):
            if t > 0:
                h = mu + phi * (h - mu) + sigma * rng.standard_normal()
            out[t, 0] = math.exp(0.5 * h) * rng.standard_normal()
    elif model.model_id == MODEL_OBJECT_MOTION:
        # README.md:87-91: still for the first half, moving for the second
        y = 0.0
        for t in range(1, T + 1):
            moving = t > T // 2
            y = y + (math.sin(t) if moving else 0.0) + model.info["sy"] * rng.standard_normal()
            out[t - 1] = (y + model.info["sobs"] * rng.standard_normal(), math.sin(t))
    elif model.model_id == MODEL_LINE:
        slope = float(rng.integers(int(P[8]), int(P[8] + P[9])))
        for t in range(1, T + 1):
            out_t = rng.random() < P[0]
            out[t - 1] = (t * slope + (model.info["s_out"] if out_t else model.info["s_in"]) * rng.standard_normal(), float(t))
    else:
        raise ValueError("unknown model")
    return out


def kalman_loglik(model: NativeModel, ys: np.ndarray) -> float:
    """Exact log p(y_1:T) of the linear-Gaussian SSM (known answer for the particle log-ML estimate)."""
    assert model.model_id == MODEL_LGSSM2
    A, sq, sr, s0 = model.info["A"], model.info["sq"], model.info["sr"], model.info["s0"]
    m, Pm = np.zeros(2), (s0 ** 2) * np.eye(2)
    Q, R = (sq ** 2) * np.eye(2), (sr ** 2) * np.eye(2)
    ll = 0.0
    for t in range(ys.shape[0]):
        if t > 0:
            m, Pm = A @ m, A @ Pm @ A.T + Q
        S = Pm + R
        e = ys[t] - m
        ll += -0.5 * (e @ np.linalg.solve(S, e) + math.log(np.linalg.det(S)) + 2.0 * math.log(2.0 * math.pi))
        Kg = Pm @ np.linalg.inv(S)
        m, Pm = m + Kg @ e, (np.eye(2) - Kg) @ Pm
    return float(ll)
