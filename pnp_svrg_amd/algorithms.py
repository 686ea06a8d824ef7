"""Host-side mirror of the reference's `algorithms/*` interface: pnp_gd / pnp_sgd / pnp_svrg /
pnp_saga / pnp_sarah (+ tune_* hyperopt objectives), driving the MI355X kernels.

Signatures, result-dict keys, logging order, stopping rules and the v1 quirks are those of
reference algorithms/pnp_gd.py:8-84, pnp_sgd.py:8-84, pnp_svrg.py:8-105 (v = mu),
pnp_saga.py:8-102, pnp_sarah.py:8-129.  The loops are wall-clock
bounded like the reference: every `time.time()` of the reference corresponds to one `clock()`
here, so a counting clock (keyword-only extension `clock=`) pins the iteration count for parity
tests.  Other extensions: `variant='svrg'` on pnp_svrg selects the true SVRG direction
(the line commented out at pnp_svrg.py:53).

With a native problem (pnp_svrg_amd.problems) and a native denoiser (pnp_svrg_amd.denoisers)
the iterate never leaves HBM: per inner iteration the host only draws the minibatch from the
global `np.random` stream (seed parity) and reads back one float64 (the squared error that the
PSNR-based stopping rule needs).  Foreign (NumPy-protocol) problems or denoisers are driven
through their own methods; the noise estimate still runs on the device.
"""
import time
import numpy as np
import torch

from . import ops

tol = 1e-5


class CountingClock:
    """Deterministic stand-in for time.time(): returns 0, 1, 2, ... (one tick per call).  Passing one as `clock=` fixes
    the iteration count of a loop (the reference loops are wall-clock bounded) and, because the schedule then does not
    depend on the device, lets pnp_svrg replay whole outer iterations as hipGraphs (`graph=`)."""
    deterministic = True

    def __init__(self):
        self.n = -1.0

    def __call__(self):
        self.n += 1.0
        return self.n

    time = __call__


# ------------------------------------------------------------------------------------------
# small adaptor layer: vectors are device tensors for native problems, NumPy arrays otherwise
# ------------------------------------------------------------------------------------------
def _native_problem(p):
    return hasattr(p, 'to_device') and hasattr(p, 'sse_device')


def _native_denoiser(d):
    return hasattr(d, 'denoise_device')


def _device_sync():
    """torch.cuda.synchronize() without its 8 us of Python per call (lazy-init and availability checks, environment lookups):
    the loops call the clock seven times per inner iteration."""
    try:
        torch._C._cuda_synchronize()
    except AttributeError:
        torch.cuda.synchronize()


class _Ctx:
    def __init__(self, problem, denoiser, clock):
        self.p, self.d = problem, denoiser
        self.native = _native_problem(problem)
        self.H, self.W = problem.H, problem.W
        self._clock = clock if clock is not None else time.time
        self._real = clock is None
        self.time_per_iter, self.psnr_per_iter = [], []
        self.gradient_time = 0
        self.denoise_time = 0

    def clock(self):
        if self._real and self.native:
            _device_sync()                      # attribute device time to the phase that queued it
        return self._clock()

    def init(self):
        if self.native:
            return self.p.to_device(self.p.Xinit).clone()
        return np.copy(self.p.Xinit)

    def copy(self, v):
        return v.clone() if isinstance(v, torch.Tensor) else np.copy(v)

    def host(self, v):
        return v.double().cpu().numpy().ravel() if isinstance(v, torch.Tensor) else v

    def psnr(self, v):
        return self.p.PSNR(v)

    def psnr_of_current(self, z):
        """problem.PSNR(z) at the top of an iteration of pnp_gd / sgd / saga / svrg (`start_PSNR`, e.g. pnp_svrg.py:50): in
        those loops z is, at that point, exactly the image whose PSNR was logged last (the initial entry, or the prox
        output of the previous iteration), so on the device path the logged value is taken instead of another error-sum
        kernel and a blocking read-back per iteration.  (Not for pnp_sarah, whose log also holds w_next's PSNR, F6.)"""
        if self.native and self.psnr_per_iter:
            return self.psnr_per_iter[-1]
        return self.psnr(z)

    def step(self, z, lr, v):
        """z -= lr*v (in place)."""
        if isinstance(z, torch.Tensor):
            ops.axpbypcz(1.0, z, -lr, v, out=z)
        else:
            z -= lr * v
        return z

    def lincomb(self, a, x, b, y, c=0.0, w=None):
        if isinstance(x, torch.Tensor):
            return ops.axpbypcz(a, x, b, y, c, w)
        out = a * x + b * y
        return out if w is None else out + c * w

    def grad_stoch(self, z, mb, n):
        """problem.grad_stoch(z, mb) / n  (the division happens inside the kernel epilogue when native)."""
        if self.native:
            return self.p.grad_stoch(z, mb, scale=1.0 / n)
        return self.p.grad_stoch(z, mb) / n

    def prox(self, z):
        """estimate_sigma + denoise; returns (denoised vector, PSNR of it)."""
        H, W = self.H, self.W
        if _native_denoiser(self.d):
            zt = z if isinstance(z, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(z)).to('cuda', self._dn_dtype())
            xrec = self.p._xrec_d if self.native else None
            out, sse, _ = self.d.denoise_device(zt.reshape(1, H, W), sigma_est=None, xrec=xrec)
            if self.native:
                return out.reshape(-1), self.p.psnr_from_sse(sse.item(), self.p.N)
            z0 = out.reshape(H, W).double().cpu().numpy()
            return np.copy(z0).ravel(), self.p.PSNR(z0)
        # foreign denoiser: NumPy protocol, noise estimate on the device
        zt = z if isinstance(z, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(z)).to('cuda', self._dn_dtype())
        sigma_est = float(ops.sigma_est(zt.reshape(1, H, W)).item())
        z0 = self.d.denoise(noisy=np.copy(self.host(z)).reshape(H, W), sigma_est=sigma_est)
        if self.native:
            zd = self.p.to_device(z0).reshape(-1)
            return zd, self.p.PSNR(zd)
        return np.copy(z0).ravel(), self.p.PSNR(z0)

    def _dn_dtype(self):
        from .problems import get_default_dtype
        return getattr(self.d, 'dtype', None) or get_default_dtype()

    def result(self, z, name):
        return {'z': self.host(z), 'time_per_iter': self.time_per_iter, 'psnr_per_iter': self.psnr_per_iter,
                'gradient_time': self.gradient_time, 'denoise_time': self.denoise_time, 'algo_name': name}

    def stop(self, start_psnr, converge_check, diverge_check):
        if converge_check is True and np.abs(start_psnr - self.psnr_per_iter[-1]) < tol:
            return True
        return diverge_check is True and self.psnr_per_iter[-1] < 0


def _run_flat(c, z, eta, tt, lr_decay, converge_check, diverge_check, verbose, direction, gd_timing, elapsed):
    """Body shared by pnp_gd / pnp_sgd / pnp_saga: one gradient step + one prox per pass."""
    i = 0
    while (c.clock() - elapsed) < tt:
        start_psnr = c.psnr_of_current(z)
        g0 = c.clock()
        v = direction(z)
        c.step(z, eta * lr_decay ** i, v)
        ge = c.clock() - g0
        c.gradient_time += ge
        if verbose:
            print(str(i) + " Before denoising:  " + str(c.psnr(z)))
        d0 = c.clock()
        z0, ps = c.prox(z)
        de = c.clock() - d0
        c.denoise_time += de
        c.time_per_iter.append((c.clock() - g0) if gd_timing else (ge + de))
        c.psnr_per_iter.append(ps)
        z = z0
        if verbose:
            print(str(i) + " After denoising:  " + str(ps))
        i += 1
        if c.stop(start_psnr, converge_check, diverge_check):
            break
    return z


def pnp_gd(problem, denoiser, eta, tt, verbose=True, lr_decay=1, converge_check=True, diverge_check=False, *, clock=None):
    c = _Ctx(problem, denoiser, clock)
    z = c.init()
    elapsed = c.clock()
    c.time_per_iter.append(c.clock() - elapsed)
    c.psnr_per_iter.append(c.psnr(z))
    z = _run_flat(c, z, eta, tt, lr_decay, converge_check, diverge_check, verbose, problem.grad_full, True, elapsed)
    return c.result(z, 'PnP GD')


def pnp_sgd(problem, denoiser, eta, tt, mini_batch_size, verbose=True, lr_decay=1, converge_check=True,
            diverge_check=False, *, clock=None):
    c = _Ctx(problem, denoiser, clock)
    z = c.init()
    elapsed = c.clock()
    c.time_per_iter.append(c.clock() - elapsed)
    c.psnr_per_iter.append(c.psnr(z))

    def direction(zz):
        mb = problem.select_mb(mini_batch_size)
        return c.grad_stoch(zz, mb, mini_batch_size)

    z = _run_flat(c, z, eta, tt, lr_decay, converge_check, diverge_check, verbose, direction, False, elapsed)
    return c.result(z, 'PnP SGD')


def pnp_saga(problem, denoiser, eta, tt, mini_batch_size, hist_size=50, verbose=True, lr_decay=1,
             converge_check=True, diverge_check=False, *, clock=None):
    c = _Ctx(problem, denoiser, clock)
    z = c.init()
    elapsed = c.clock()
    t0 = c.clock()
    mb = problem.select_mb(mini_batch_size)
    g0 = c.grad_stoch(z, mb, mini_batch_size)
    # reference: table = hist_size aliases of one gradient; v uses sum(table)/hist_size every step.
    # Here: the table lives in HBM, and its sum is kept incrementally (algebraically equal).
    table = [g0] * hist_size
    tsum = c.lincomb(float(hist_size), g0, 0.0, g0)
    state = {'prev': g0, 'sum': tsum}
    c.time_per_iter.append(c.clock() - t0)
    c.psnr_per_iter.append(c.psnr(z))

    def direction(zz):
        mbb = problem.select_mb(mini_batch_size)
        r = np.random.choice(hist_size, 1).item()
        g = c.grad_stoch(zz, mbb, mini_batch_size)
        state['sum'] = c.lincomb(1.0, state['sum'], 1.0, g, -1.0, table[r])
        table[r] = g
        v = c.lincomb(1.0, g, -1.0, state['prev'], 1.0 / hist_size, state['sum'])
        state['prev'] = g
        return v

    z = _run_flat(c, z, eta, tt, lr_decay, converge_check, diverge_check, verbose, direction, False, elapsed)
    return c.result(z, 'pnp_saga')


class _SvrgGraph:
    """One outer iteration of pnp_svrg on a native CSMRI problem (B = 1) as a hipGraph: full gradient, w = z, PSNR log,
    T2 x (selector from the host-drawn index list, SVRG step, prox, PSNR log).  Minibatch index lists live in a device
    buffer that the host refills before every replay (they come from the legacy np.random stream); squared errors go to
    a device log that is read back once at the end.  At B = 1 an inner iteration is ~8 small kernels, so launch
    latency is the whole cost: a replay removes it."""

    def __init__(self, problem, denoiser, eta, T2, mb, variant, n_log):
        p = self.p = problem
        self.d, self.eta, self.T2, self.mb, self.variant = denoiser, eta, T2, mb, variant
        dev, H, W = p.device, p.H, p.W
        self.z = p.to_device(p.Xinit).clone().reshape(1, H, W)
        self.w = torch.empty_like(self.z)
        self.mu = torch.empty_like(self.z)
        self.idx = torch.zeros((T2, 1, mb), dtype=torch.int32, device=dev)
        self.selT = torch.empty_like(p._maskT)
        self.sse = torch.zeros(1, dtype=torch.float64, device=dev)
        self.log = torch.zeros((n_log, 1), dtype=torch.float64, device=dev)
        self.cnt = torch.zeros(1, dtype=torch.int32, device=dev)
        self.graph = None

    def log_psnr(self):
        ops.sse(self.z, self.p._xrec_d, out=self.sse)
        ops.log_append_inc(self.sse, self.log, self.cnt)

    def refresh(self):
        p = self.p
        p.plan.grad(self.z, p._maskT, yh=p._yh_full, alpha=1.0 / p.M0, out=self.mu)
        self.w.copy_(self.z)
        self.log_psnr()

    def inner(self, j):
        p, lr = self.p, self.eta
        if self.variant == 'svrg':
            p.plan.sel_from_indices(self.idx[j], out=self.selT)
            p.plan.grad(self.z, self.selT, b=self.w, alpha=-lr / self.mb, beta=1.0, c1=self.z, gamma=-lr, c2=self.mu, out=self.z)
        else:
            ops.axpbypcz(1.0, self.z, -lr, self.mu, out=self.z)
        self.d.prox_inplace(self.z, self.p._xrec_d, self.sse)
        ops.log_append_inc(self.sse, self.log, self.cnt)

    def outer_body(self):
        self.refresh()
        for j in range(self.T2):
            self.inner(j)

    def capture(self):
        keep = (self.z.clone(), self.log.clone(), self.cnt.clone())
        t0 = getattr(self.d, 't', 0)
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            self.outer_body()                                   # warm-up outside capture (lazy module loads)
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        self.t_per_outer = getattr(self.d, 't', 0) - t0          # the denoiser's call counter advances per prox (TV.py:22)
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            self.outer_body()
        torch.cuda.synchronize()
        for dst, src in zip((self.z, self.log, self.cnt), keep):
            dst.copy_(src)
        if hasattr(self.d, 't'):
            self.d.t = t0
        self.graph = g

    def upload(self, outers):
        """all minibatch index lists of the run in ONE host-to-device copy: [outer][T2][mb] (rows of a short last outer
        iteration stay zero and are never read)"""
        buf = np.full((max(len(outers), 1), self.T2, self.mb), -1, np.int32)      # -1: no location (the scatter kernel skips it)
        for o, lists in enumerate(outers):
            for j, l in enumerate(lists):
                buf[o, j, :len(l)] = l
        self.all_idx = torch.from_numpy(buf).to(self.p.device)

    def run_outer(self, o, n):
        """outer iteration o with n <= T2 inner iterations (a short last one runs eagerly)"""
        self.idx.copy_(self.all_idx[o].reshape(self.T2, 1, self.mb))     # device-to-device, stream-ordered
        if n == self.T2:
            if self.graph is None:
                self.capture()
            self.graph.replay()
            if hasattr(self.d, 't'):
                self.d.t += self.t_per_outer
        else:
            self.refresh()
            for j in range(n):
                self.inner(j)


def _svrg_graph_schedule(c, problem, tt, T2, mini_batch_size):
    """Dry run of pnp_svrg's loop nest against a deterministic clock: the same clock() calls in the same order, the same
    select_mb draws from the legacy np.random stream, the same time bookkeeping -- and no device work.  Returns the
    minibatch index lists per outer iteration."""
    # a select_mb that is the class's own (not overridden / monkeypatched) can skip building the H x W indicator
    fast = (hasattr(problem, '_select_mb_locs') and 'select_mb' not in problem.__dict__
            and type(problem).select_mb.__qualname__ == 'CSMRI.select_mb')
    elapsed = c.clock()
    c.time_per_iter.append(c.clock() - elapsed)
    outers = []
    while (c.clock() - elapsed) < tt:
        start_time = c.clock()
        c.time_per_iter.append(c.clock() - start_time)
        lists = []
        for j in range(T2):
            if (c.clock() - elapsed) >= tt:
                break
            g0 = c.clock()
            if fast:
                locs = problem._select_mb_locs(mini_batch_size)
            else:
                mb = problem.select_mb(mini_batch_size)
                locs = np.flatnonzero(np.multiply(problem.mask, np.asarray(mb).reshape(problem.H, problem.W)))
            ge = c.clock() - g0
            c.gradient_time += ge
            d0 = c.clock()
            de = c.clock() - d0
            c.denoise_time += de
            c.time_per_iter.append(ge + de)
            lists.append(locs)
        outers.append(lists)
    return outers


def _svrg_graph_eligible(c, problem, denoiser, clock, lr_decay, verbose, converge_check, diverge_check):
    # (a select_mb that is not the class's own -- overridden, monkeypatched, fed index lists by a test -- may return any number
    #  of locations, also outside the mask: only the eager loop reproduces what the reference does with those)
    own_select = ('select_mb' not in problem.__dict__ and getattr(type(problem).select_mb, '__qualname__', '') == 'CSMRI.select_mb')
    return (c.native and getattr(problem, 'pname', '') == 'csmri' and hasattr(problem, 'plan') and own_select and
            getattr(clock, 'deterministic', False) and lr_decay == 1 and not verbose and converge_check is not True
            and diverge_check is not True and hasattr(denoiser, 'prox_inplace') and denoiser.prox_inplace(None, None, None, probe=True))


def pnp_svrg(problem, denoiser, eta, tt, T2, mini_batch_size, verbose=True, lr_decay=1, converge_check=True,
             diverge_check=False, *, clock=None, variant='reference', graph=None):
    """graph (extension): None = replay whole outer iterations as hipGraphs when nothing in the loop depends on the
    device -- a deterministic `clock` (CountingClock), no convergence / divergence test, lr_decay == 1, not verbose, a
    native CSMRI problem and an in-place native prox; False = never.  Same results either way."""
    if variant not in ('reference', 'svrg'):
        raise ValueError("variant must be 'reference' (v = mu, what the reference executes) or 'svrg'")
    c = _Ctx(problem, denoiser, clock)
    if graph is not False and _svrg_graph_eligible(c, problem, denoiser, clock, lr_decay, verbose, converge_check, diverge_check):
        outers = _svrg_graph_schedule(c, problem, tt, T2, mini_batch_size)
        n_log = 1 + sum(1 + len(l) for l in outers)
        run = _SvrgGraph(problem, denoiser, eta, T2, mini_batch_size, variant, n_log)
        run.upload(outers)
        run.log_psnr()                                          # the initial entry
        for o, lists in enumerate(outers):
            run.run_outer(o, len(lists))
        sse = run.log[:, 0].cpu().numpy()
        c.psnr_per_iter = [problem.psnr_from_sse(v, problem.N) for v in sse]
        return c.result(run.z.reshape(-1), 'PnP SVRG')
    z = c.init()
    fused = variant == 'svrg' and c.native and hasattr(problem, 'grad_stoch_diff')
    i = 0
    elapsed = c.clock()
    c.time_per_iter.append(c.clock() - elapsed)
    c.psnr_per_iter.append(c.psnr(z))
    break_out_flag = False
    while (c.clock() - elapsed) < tt:
        if break_out_flag:
            break
        start_time = c.clock()
        mu = problem.grad_full(z)
        w = c.copy(z)
        c.time_per_iter.append(c.clock() - start_time)
        c.psnr_per_iter.append(c.psnr(z))
        for j in range(T2):
            if (c.clock() - elapsed) >= tt:
                break
            start_psnr = c.psnr_of_current(z)
            g0 = c.clock()
            mb = problem.select_mb(mini_batch_size)          # drawn even when unused (RNG parity, F1)
            lr = eta * lr_decay ** i
            if fused:
                # z <- z - lr*((gs(z,mb) - gs(w,mb))/mb + mu): one FFT pair, fused epilogue
                problem.grad_stoch_diff(z, w, mb, alpha=-lr / mini_batch_size, beta=1.0, c1=z, gamma=-lr, c2=mu, out=z)
            elif variant == 'svrg':
                v = c.lincomb(1.0, c.grad_stoch(z, mb, mini_batch_size), -1.0, c.grad_stoch(w, mb, mini_batch_size), 1.0, mu)
                c.step(z, lr, v)
            else:
                c.step(z, lr, mu)
            ge = c.clock() - g0
            c.gradient_time += ge
            if verbose:
                print(str(i) + " " + str(j) + " Before denoising:  " + str(c.psnr(z)))
            d0 = c.clock()
            z0, ps = c.prox(z)
            de = c.clock() - d0
            c.denoise_time += de
            c.time_per_iter.append(ge + de)
            c.psnr_per_iter.append(ps)
            z = z0
            if verbose:
                print("After denoising update: " + str(i) + " " + str(j) + " " + str(ps))
            if c.stop(start_psnr, converge_check, diverge_check):
                break_out_flag = True
                break
        i += 1
    return c.result(z, 'PnP SVRG')


def pnp_sarah(problem, denoiser, eta, tt, T2, mini_batch_size, verbose=True, lr_decay=1, converge_check=True,
              diverge_check=False, *, clock=None):
    c = _Ctx(problem, denoiser, clock)
    z = c.init()
    i = 0
    elapsed = c.clock()
    break_out_flag = False
    while (c.clock() - elapsed) < tt:
        if break_out_flag:
            break
        w_previous = c.copy(z)
        g0 = c.clock()
        v_previous = problem.grad_full(z)
        w_next = c.lincomb(1.0, w_previous, -eta, v_previous)       # no lr_decay here (F6c)
        ge = c.clock() - g0
        c.gradient_time += ge
        d0 = c.clock()
        w_next, ps = c.prox(w_next)
        de = c.clock() - d0
        c.denoise_time += de
        c.time_per_iter.append(ge + de)
        c.psnr_per_iter.append(ps)
        for j in range(T2):
            if (c.clock() - elapsed) >= tt:
                break
            start_psnr = c.psnr(z)
            g0 = c.clock()
            mb = problem.select_mb(mini_batch_size)
            if c.native and hasattr(problem, 'grad_stoch_diff'):
                v_next = problem.grad_stoch_diff(w_next, w_previous, mb, alpha=1.0 / mini_batch_size, beta=1.0,
                                                 c1=v_previous).reshape(-1)
            else:
                v_next = c.lincomb(1.0, c.grad_stoch(w_next, mb, mini_batch_size), -1.0,
                                   c.grad_stoch(w_previous, mb, mini_batch_size), 1.0, v_previous)
            c.step(z, eta * lr_decay ** i, v_next)
            ge = c.clock() - g0
            c.gradient_time += ge
            if verbose:
                print("After gradient update: " + str(i) + " " + str(j) + " " + str(c.psnr(z)))
            d0 = c.clock()
            z0, ps = c.prox(z)
            de = c.clock() - d0
            c.denoise_time += de
            v_previous = v_next
            w_previous = c.copy(z0)
            c.time_per_iter.append(ge + de)
            c.psnr_per_iter.append(ps)
            z = z0
            if verbose:
                print("After denoising update: " + str(i) + " " + str(j) + " " + str(ps))
            if c.stop(start_psnr, converge_check, diverge_check):
                break_out_flag = True
                break
        i += 1
    return c.result(z, 'pnp_sarah')


# ------------------------------------------------------------------------------------------
# hyperopt objectives (reference algorithms/pnp_*.py tune_* wrappers)
# ------------------------------------------------------------------------------------------
def _tuned(result, problem):
    try:
        from hyperopt import STATUS_OK
    except ImportError:                        # hyperopt is optional; its constant is the string 'ok'
        STATUS_OK = 'ok'
    return {'loss': (problem.PSNR(problem.Xinit) - problem.PSNR(result['z'])), 'status': STATUS_OK,
            'algo_name': result['algo_name'], 'z': result['z'], 'time_per_iter': result['time_per_iter'],
            'psnr_per_iter': result['psnr_per_iter'], 'gradient_time': result['gradient_time'],
            'denoise_time': result['denoise_time']}


def tune_pnp_gd(args, problem, denoiser, tt, lr_decay=1, verbose=False, converge_check=True, diverge_check=True):
    eta, dstrength = args
    denoiser.sigma_est = dstrength             # an attribute no denoiser reads (F14); kept for parity
    return _tuned(pnp_gd(eta=eta, problem=problem, denoiser=denoiser, tt=tt, verbose=verbose, lr_decay=lr_decay,
                         converge_check=converge_check, diverge_check=diverge_check), problem)


def tune_pnp_sgd(args, problem, denoiser, tt, lr_decay=1, verbose=False, converge_check=True, diverge_check=True):
    eta, mini_batch_size, dstrength = args
    denoiser.sigma_est = dstrength
    return _tuned(pnp_sgd(eta=eta, mini_batch_size=mini_batch_size, problem=problem, denoiser=denoiser, tt=tt,
                          verbose=verbose, lr_decay=lr_decay, converge_check=converge_check,
                          diverge_check=diverge_check), problem)


def tune_pnp_svrg(args, problem, denoiser, tt, lr_decay=1, verbose=False, converge_check=True, diverge_check=True):
    eta, mini_batch_size, T2, dstrength = args
    denoiser.sigma_est = dstrength
    return _tuned(pnp_svrg(eta=eta, mini_batch_size=mini_batch_size, T2=T2, problem=problem, denoiser=denoiser, tt=tt,
                           verbose=verbose, lr_decay=lr_decay, converge_check=converge_check,
                           diverge_check=diverge_check), problem)


def tune_pnp_saga(args, problem, denoiser, tt, lr_decay=1, verbose=False, converge_check=True, diverge_check=True):
    eta, mini_batch_size, dstrength, hist_size = args
    denoiser.sigma_est = dstrength
    return _tuned(pnp_saga(eta=eta, mini_batch_size=mini_batch_size, hist_size=hist_size, problem=problem,
                           denoiser=denoiser, tt=tt, verbose=verbose, lr_decay=lr_decay,
                           converge_check=converge_check, diverge_check=diverge_check), problem)


def tune_pnp_sarah(args, problem, denoiser, tt, lr_decay=1, verbose=False, converge_check=True, diverge_check=True):
    eta, mini_batch_size, T2, dstrength = args
    denoiser.sigma_est = dstrength
    return _tuned(pnp_sarah(eta=eta, mini_batch_size=mini_batch_size, T2=T2, problem=problem, denoiser=denoiser,
                            tt=tt, verbose=verbose, lr_decay=lr_decay, converge_check=converge_check,
                            diverge_check=diverge_check), problem)
