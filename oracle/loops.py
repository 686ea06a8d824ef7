"""Oracle (test infrastructure): the five PnP loops, restated.

Follows reference algorithms/pnp_gd.py:8-84, pnp_sgd.py:8-84, pnp_svrg.py:8-105,
pnp_saga.py:8-102, pnp_sarah.py:8-129 including their v1 quirks:
the loops are wall-clock bounded, so every `time.time()` call of the reference is
mirrored by one `clock()` call here, in the same order -- a counting fake clock then
fixes the iteration count identically in the reference, here and in the product.
"""
import time
import numpy as np
from . import denoise as _dn

TOL = 1e-5


class CountingClock:
    """Deterministic stand-in for time.time(): returns 0, 1, 2, ... (one tick per call)."""
    deterministic = True            # (the product's loops may pre-compute their schedule from such a clock)

    def __init__(self):
        self.n = -1.0

    def __call__(self):
        self.n += 1.0
        return self.n

    time = __call__


class _Log:
    def __init__(self, name):
        self.time_per_iter, self.psnr_per_iter = [], []
        self.gradient_time = self.denoise_time = 0
        self.name = name

    def result(self, z):
        return {'z': z, 'time_per_iter': self.time_per_iter, 'psnr_per_iter': self.psnr_per_iter,
                'gradient_time': self.gradient_time, 'denoise_time': self.denoise_time,
                'algo_name': self.name}


def _prox(problem, denoiser, z):
    z0 = np.copy(z).reshape(problem.H, problem.W)
    return denoiser.denoise(noisy=z0, sigma_est=_dn.estimate_sigma(z0))


def _stop(log, start_psnr, converge_check, diverge_check):
    if converge_check is True and np.abs(start_psnr - log.psnr_per_iter[-1]) < TOL:
        return True
    return diverge_check is True and log.psnr_per_iter[-1] < 0


def _single_loop(problem, denoiser, eta, tt, lr_decay, converge_check, diverge_check, clock,
                 log, z, direction, total_time):
    """Shared body of pnp_gd / pnp_sgd / pnp_saga after their prologues."""
    i = 0
    elapsed = log.elapsed
    while (clock() - elapsed) < tt:
        start_psnr = problem.PSNR(z)
        g0 = clock()
        v = direction(z)
        z -= (eta * lr_decay ** i) * v
        ge = clock() - g0
        log.gradient_time += ge
        d0 = clock()
        z0 = _prox(problem, denoiser, z)
        de = clock() - d0
        log.denoise_time += de
        log.time_per_iter.append((clock() - g0) if total_time else (ge + de))
        log.psnr_per_iter.append(problem.PSNR(z0))
        z = np.copy(z0).ravel()
        i += 1
        if _stop(log, start_psnr, converge_check, diverge_check):
            break
    return z


def pnp_gd(problem, denoiser, eta, tt, verbose=False, lr_decay=1, converge_check=True,
           diverge_check=False, clock=time.time):
    log = _Log('PnP GD')
    z = np.copy(problem.Xinit)
    log.elapsed = clock()
    log.time_per_iter.append(clock() - log.elapsed)
    log.psnr_per_iter.append(problem.PSNR(z))
    z = _single_loop(problem, denoiser, eta, tt, lr_decay, converge_check, diverge_check, clock,
                     log, z, problem.grad_full, total_time=True)        # pnp_gd.py:58
    return log.result(z)


def pnp_sgd(problem, denoiser, eta, tt, mini_batch_size, verbose=False, lr_decay=1,
            converge_check=True, diverge_check=False, clock=time.time):
    log = _Log('PnP SGD')
    z = np.copy(problem.Xinit)
    log.elapsed = clock()
    log.time_per_iter.append(clock() - log.elapsed)
    log.psnr_per_iter.append(problem.PSNR(z))

    def direction(zz):                                                   # pnp_sgd.py:32-33
        mb = problem.select_mb(mini_batch_size)
        return problem.grad_stoch(zz, mb) / mini_batch_size

    z = _single_loop(problem, denoiser, eta, tt, lr_decay, converge_check, diverge_check, clock,
                     log, z, direction, total_time=False)
    return log.result(z)


def pnp_saga(problem, denoiser, eta, tt, mini_batch_size, hist_size=50, verbose=False, lr_decay=1,
             converge_check=True, diverge_check=False, clock=time.time):
    log = _Log('pnp_saga')
    z = np.copy(problem.Xinit)
    log.elapsed = clock()
    t0 = clock()
    mb = problem.select_mb(mini_batch_size)                              # pnp_saga.py:25-29
    g0 = problem.grad_stoch(z, mb) / mini_batch_size
    table = [g0] * hist_size
    state = {'prev': g0}
    log.time_per_iter.append(clock() - t0)
    log.psnr_per_iter.append(problem.PSNR(z))

    def direction(zz):                                                   # pnp_saga.py:43-47,72
        mbb = problem.select_mb(mini_batch_size)
        r = np.random.choice(hist_size, 1).item()
        table[r] = problem.grad_stoch(zz, mbb) / mini_batch_size
        v = table[r].ravel() - state['prev'].ravel() + sum(table).ravel() / hist_size
        state['prev'] = table[r]
        return v

    z = _single_loop(problem, denoiser, eta, tt, lr_decay, converge_check, diverge_check, clock,
                     log, z, direction, total_time=False)
    return log.result(z)


def pnp_svrg(problem, denoiser, eta, tt, T2, mini_batch_size, verbose=False, lr_decay=1,
             converge_check=True, diverge_check=False, clock=time.time, variant='reference'):
    """variant='reference': v = mu (what v1 executes, pnp_svrg.py:54);
    variant='svrg': the commented-out line pnp_svrg.py:53."""
    log = _Log('PnP SVRG')
    z = np.copy(problem.Xinit)
    i = 0
    elapsed = clock()
    log.time_per_iter.append(clock() - elapsed)
    log.psnr_per_iter.append(problem.PSNR(z))
    done = False
    while (clock() - elapsed) < tt:
        if done:
            break
        t0 = clock()
        mu = problem.grad_full(z)
        w = np.copy(z)
        log.time_per_iter.append(clock() - t0)
        log.psnr_per_iter.append(problem.PSNR(z))
        for _ in range(T2):
            if (clock() - elapsed) >= tt:
                break
            start_psnr = problem.PSNR(z)
            g0 = clock()
            mb = problem.select_mb(mini_batch_size)
            if variant == 'svrg':
                v = (problem.grad_stoch(z, mb) - problem.grad_stoch(w, mb)) / mini_batch_size + mu
            else:
                v = mu
            z -= (eta * lr_decay ** i) * v
            ge = clock() - g0
            log.gradient_time += ge
            d0 = clock()
            z0 = _prox(problem, denoiser, z)
            de = clock() - d0
            log.denoise_time += de
            log.time_per_iter.append(ge + de)
            log.psnr_per_iter.append(problem.PSNR(z0))
            z = np.copy(z0).ravel()
            if _stop(log, start_psnr, converge_check, diverge_check):
                done = True
                break
        i += 1
    return log.result(z)


def pnp_sarah(problem, denoiser, eta, tt, T2, mini_batch_size, verbose=False, lr_decay=1,
              converge_check=True, diverge_check=False, clock=time.time):
    log = _Log('pnp_sarah')
    z = np.copy(problem.Xinit)
    i = 0
    elapsed = clock()
    done = False
    while (clock() - elapsed) < tt:
        if done:
            break
        w_prev = np.copy(z)
        g0 = clock()
        v_prev = problem.grad_full(z)
        w_next = w_prev - eta * v_prev                                   # pnp_sarah.py:35 (no decay)
        ge = clock() - g0
        log.gradient_time += ge
        d0 = clock()
        w_next = _prox(problem, denoiser, w_next)
        de = clock() - d0
        log.denoise_time += de
        log.time_per_iter.append(ge + de)
        log.psnr_per_iter.append(problem.PSNR(w_next))
        w_next = w_next.ravel()
        for _ in range(T2):
            if (clock() - elapsed) >= tt:
                break
            start_psnr = problem.PSNR(z)
            g0 = clock()
            mb = problem.select_mb(mini_batch_size)
            v_next = (problem.grad_stoch(w_next, mb).ravel()
                      - problem.grad_stoch(w_prev, mb).ravel()) / mini_batch_size + v_prev.ravel()
            z -= (eta * lr_decay ** i) * v_next
            ge = clock() - g0
            log.gradient_time += ge
            d0 = clock()
            z0 = _prox(problem, denoiser, z)
            de = clock() - d0
            log.denoise_time += de
            v_prev = np.copy(v_next)
            w_prev = np.copy(z0).ravel()
            log.time_per_iter.append(ge + de)
            log.psnr_per_iter.append(problem.PSNR(z0))
            z = np.copy(z0).ravel()
            if _stop(log, start_psnr, converge_check, diverge_check):
                done = True
                break
        i += 1
    return log.result(z)
