"""CPU restatement of the reference's screen simulator -- TEST INFRASTRUCTURE (input generator).

``scint_sim.Simulation`` (/root/reference/scintools/scint_sim.py:23-311, after Coles et al. 2010)
is how the reference makes its test inputs; SURVEY.md 8(d) quotes the BASELINE configs on it
(``mb2=20, ar=10, psi=0, alpha=5/3, inner=0.001, ds=0.01, dlam=0.25, freq=1400, dt=30``).  The
reference cannot be imported on the GPU box, so the part of it that produces the dynamic
spectrum is restated here, operation by operation (same NumPy calls, same order, same
complex64 storage of the field), which makes the output BIT-IDENTICAL to the reference's for a
given seed.  Only ``tests/`` and ``bench.py --input simulation`` (input creation, outside the
timed region) may import this module; the product never does.

PARITY PIN: ``tests/test_oracle_golden.py::test_sim_oracle_*`` compares the arrays with the
reference's own ``Simulation`` output stored in ``tests/golden/sim_sspec.npz`` (full 128 x 96
array) and with checksums of the 1024^2 / 2048^2 runs in ``tests/golden/sim_sweep.npz``
(``tests/golden/make_golden.py::gen_sim_sweep`` ran the unmodified reference).

Not restated: the impulse response (``get_pulse``), plots, ``lamsteps=True``.
"""
import numpy as np
from numpy.fft import fft2, ifft2
from scipy.special import gamma


class Simulation:
    """Attributes as the reference sets them: dyn[nf, nx] (float32), freqs [MHz], times [s],
    dt, df, eta [s^3], plus the screen (xyp) and the field spectrum (spe)."""

    def __init__(self, mb2=2, rf=1, ds=0.01, alpha=5 / 3, ar=1, psi=0, inner=0.001, ns=256, nf=256,
                 dlam=0.25, seed=None, nx=None, ny=None, dx=None, dy=None, freq=1400, dt=30, mjd=60000, workers=1):
        self.workers = int(workers)
        self.mb2, self.rf, self.ds, self.alpha, self.ar, self.psi, self.inner = mb2, rf, ds, alpha, ar, psi, inner
        self.dx = dx if dx is not None else ds
        self.dy = dy if dy is not None else ds
        self.nx = nx if nx is not None else ns
        self.ny = ny if ny is not None else ns
        self.nf, self.dlam, self.seed = nf, dlam, seed
        self._set_constants()
        self._get_screen()
        self._get_intensity()
        spi = np.real(np.multiply(self.spe, np.conj(self.spe)))          # scint_sim.py:244
        # physical units (scint_sim.py:91-131, lamsteps=False)
        self.dt, self.freq, self.mjd = dt, freq, mjd
        self.nsub, self.nchan = int(np.shape(spi)[0]), int(np.shape(spi)[1])
        self.df = self.freq * self.dlam / (self.nchan - 1)
        self.freqs = self.freq + np.arange(-self.nchan / 2, self.nchan / 2, 1) * self.df
        self.bw = max(self.freqs) - min(self.freqs)
        self.times = self.dt * np.arange(0, self.nsub)
        self.df = self.bw / self.nchan
        self.tobs = float(self.times[-1] - self.times[0])
        self.dyn = np.transpose(spi)
        V = self.ds / self.dt
        k = 2 * np.pi / self.freq
        L = self.rf**2 * k
        self.eta = L / (2 * V**2) / 10**6 / np.cos(psi * np.pi / 180) ** 2
        self.name = 'sim:mb2={0},ar={1},psi={2},dlam={3}'.format(self.mb2, self.ar, self.psi, self.dlam)
        self.header = [self.name, 'MJD0: {}'.format(mjd)]

    def _set_constants(self):                                            # scint_sim.py:137-167
        ns = 1
        lenx, leny = self.nx * self.dx, self.ny * self.dy
        self.ffconx = (2.0 / (ns * lenx * lenx)) * (np.pi * self.rf) ** 2
        self.ffcony = (2.0 / (ns * leny * leny)) * (np.pi * self.rf) ** 2
        dqx, dqy = 2 * np.pi / lenx, 2 * np.pi / leny
        a2 = self.alpha * 0.5
        ab = 1.0 - a2
        cmb2 = self.alpha * self.mb2 / (4 * np.pi * gamma(ab) * np.cos(self.alpha * np.pi * 0.25) * ns)
        self.consp = cmb2 * dqx * dqy / (self.rf**self.alpha)

    def _swdsp(self, kx=0, ky=0):                                        # scint_sim.py:281-297
        cs, sn = np.cos(self.psi * np.pi / 180), np.sin(self.psi * np.pi / 180)
        r = self.ar
        con = np.sqrt(self.consp)
        alf = -(self.alpha + 2) / 4
        a = (cs**2) / r + r * sn**2
        b = r * cs**2 + sn**2 / r
        c = 2 * cs * sn * (1 / r - r)
        q2 = a * np.power(kx, 2) + b * np.power(ky, 2) + c * np.multiply(kx, ky)
        return con * np.multiply(np.power(q2, alf),
                                 np.exp(-(np.add(np.power(kx, 2), np.power(ky, 2))) * self.inner**2 / 2))

    def _get_screen(self):                                               # scint_sim.py:169-207
        np.random.seed(self.seed)
        nx, ny = self.nx, self.ny
        nx2, ny2 = int(nx / 2 + 1), int(ny / 2 + 1)
        w = np.zeros([nx, ny])
        dqx, dqy = 2 * np.pi / (self.dx * nx), 2 * np.pi / (self.dy * ny)
        k = np.arange(2, nx2 + 1)
        w[k - 1, 0] = self._swdsp(kx=(k - 1) * dqx, ky=0)
        w[nx + 1 - k, 0] = w[k, 0]
        ll = np.arange(2, ny2 + 1)
        w[0, ll - 1] = self._swdsp(kx=0, ky=(ll - 1) * dqy)
        w[0, ny + 1 - ll] = w[0, ll - 1]
        kp = np.arange(2, nx2 + 1)
        k = np.arange((nx2 + 1), nx + 1)
        km = -(nx - k + 1)
        for il in range(2, ny2 + 1):
            w[kp - 1, il - 1] = self._swdsp(kx=(kp - 1) * dqx, ky=(il - 1) * dqy)
            w[k - 1, il - 1] = self._swdsp(kx=km * dqx, ky=(il - 1) * dqy)
            w[nx + 1 - kp, ny + 1 - il] = w[kp - 1, il - 1]
            w[nx + 1 - k, ny + 1 - il] = w[k - 1, il - 1]
        xyp = np.multiply(w, np.add(np.random.randn(nx, ny), 1j * np.random.randn(nx, ny)))
        self.xyp = np.real(fft2(xyp))

    def _frfilt3(self, xye, scale):                                      # scint_sim.py:299-316
        nx, ny = self.nx, self.ny
        nx2, ny2 = int(nx / 2) + 1, int(ny / 2) + 1
        filt = np.zeros([nx2, ny2], dtype=np.dtype(np.csingle))
        q2x = np.linspace(0, nx2 - 1, nx2) ** 2 * scale * self.ffconx
        for ly in range(0, ny2):
            q2 = q2x + (self.ffcony * (ly**2) * scale)
            filt[:, ly] = np.cos(q2) - 1j * np.sin(q2)
        xye[0:nx2, 0:ny2] = np.multiply(xye[0:nx2, 0:ny2], filt[0:nx2, 0:ny2])
        xye[nx:nx2 - 1:-1, 0:ny2] = np.multiply(xye[nx:nx2 - 1:-1, 0:ny2], filt[1:(nx2 - 1), 0:ny2])
        xye[0:nx2, ny:ny2 - 1:-1] = np.multiply(xye[0:nx2, ny:ny2 - 1:-1], filt[0:nx2, 1:(ny2 - 1)])
        xye[nx:nx2 - 1:-1, ny:ny2 - 1:-1] = np.multiply(xye[nx:nx2 - 1:-1, ny:ny2 - 1:-1],
                                                        filt[1:(nx2 - 1), 1:(ny2 - 1)])
        return xye

    def _field_column(self, ifreq):                                      # body of the loop of scint_sim.py:215-235
        frfreq = 1.0 + self.dlam * (-0.5 + ifreq / self.nf)
        scale = 1 / frfreq
        xye = fft2(np.exp(1j * self.xyp * scale))
        xye = self._frfilt3(xye, scale)
        xye = ifft2(xye)
        return xye[:, int(np.floor(self.ny / 2))]

    def _get_intensity(self):                                            # scint_sim.py:209-236
        spe = np.zeros([self.nx, self.nf], dtype=np.dtype(np.csingle)) + \
            1j * np.zeros([self.nx, self.nf], dtype=np.dtype(np.csingle))
        import multiprocessing as _mp
        # (never from inside a worker process: a caller without a `__main__` guard is re-imported by every spawned child, and each
        #  copy would start a pool of its own)
        if self.workers > 1 and self.nf >= 4 * self.workers and _mp.parent_process() is None:
            # The frequencies are independent (each is two FFTs of the same screen): dealt to worker processes in contiguous
            # blocks, every column is the same single-threaded NumPy arithmetic as in the loop below -- the same bits
            # (tests/test_oracle_golden.py compares both with the reference's own array) -- in a fraction of the 3 minutes a
            # 4096^2 screen takes one core.  Spawned, not forked: bench.py calls this with a HIP context alive.
            import multiprocessing as mp
            bounds = np.linspace(0, self.nf, self.workers * 4 + 1).astype(int)
            jobs = [(self._worker_state(), int(a), int(b)) for a, b in zip(bounds[:-1], bounds[1:]) if b > a]
            done = np.zeros(self.nf, dtype=bool)
            try:
                with mp.get_context("spawn").Pool(self.workers) as pool:
                    it = pool.imap_unordered(_field_columns, jobs)
                    for _ in jobs:
                        # bounded wait: a pool whose workers die at start-up (a main module that cannot be re-imported)
                        # respawns them for ever -- an input generator must never hang its caller
                        a, cols = it.next(timeout=120 + 0.05 * self.nx * self.ny * (self.nf / len(jobs)) / 1e4)
                        spe[:, a:a + cols.shape[1]] = cols
                        done[a:a + cols.shape[1]] = True
            except Exception as exc:      # noqa: BLE001 -- whatever went wrong, the serial loop below still gives the same array
                import warnings
                warnings.warn(f"sim_oracle: worker pool failed ({exc!r}); {int((~done).sum())} of {self.nf} frequencies fall back to "
                              "the serial loop (same values, minutes slower)", RuntimeWarning)
            for ifreq in np.nonzero(~done)[0]:
                spe[:, ifreq] = self._field_column(int(ifreq))
        else:
            for ifreq in range(0, self.nf):
                spe[:, ifreq] = self._field_column(ifreq)
        self.spe = spe

    def _worker_state(self):
        return dict(xyp=self.xyp, nx=self.nx, ny=self.ny, nf=self.nf, dlam=self.dlam, ffconx=self.ffconx, ffcony=self.ffcony)


def _field_columns(job):
    """Worker of Simulation._get_intensity: the field columns of frequencies a .. b-1 (complex64, as the loop stores them)."""
    state, a, b = job
    sim = Simulation.__new__(Simulation)
    sim.__dict__.update(state)
    out = np.zeros([sim.nx, b - a], dtype=np.dtype(np.csingle))
    for k, ifreq in enumerate(range(a, b)):
        out[:, k] = sim._field_column(ifreq)
    return a, out


BASELINE_SCREEN = dict(mb2=20, ar=10, psi=0, alpha=5 / 3, inner=0.001, ds=0.01, dlam=0.25, freq=1400, dt=30)


def default_workers(cap=32):
    """Worker processes for the larger screens: the cores this process may use, at most `cap`."""
    import os
    try:
        n = len(os.sched_getaffinity(0))
    except (AttributeError, OSError):
        n = os.cpu_count() or 1
    return max(1, min(cap, n))


def baseline_dynspec(size, seed, ny=128, workers=1):
    """The `Simulation` input of a BASELINE config: size x size, anisotropic screen.
    `workers` > 1 deals the frequencies to that many processes (same bits, see _get_intensity)."""
    return Simulation(nx=size, nf=size, ny=ny, seed=seed, workers=workers, **BASELINE_SCREEN)


def checksum(a):
    """SHA-256 of an array's exact bytes (C order): pins a large reference run in a few bytes."""
    import hashlib
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()
