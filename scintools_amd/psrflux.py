"""psrflux dynamic-spectrum files (the format the reference reads and writes in
``Dynspec.load_file`` / ``write_file``, dynspec.py:144-230, 330-376).

Format: comment lines start with ``#`` (one of them may be ``# MJD0: <float>``), every other
line is ``isub ichan time(min) freq(MHz) flux flux_err``, sub-integration major.

This is host-side I/O in front of the GPU path, written for latency on small observations
(the survey's note: ``np.loadtxt`` dominates there):

* :func:`read_table` tokenises the whole file in one pass and converts all numbers with a
  single ``np.array(tokens, float)`` (correctly rounded, i.e. the same float64 values as
  ``np.loadtxt``);
* :func:`observation` turns the table into the attribute set the reference's ``Dynspec``
  carries (same definitions and roundings, so a file loads to identical attributes);
* :func:`save_sidecar` / :func:`load_sidecar` keep a binary ``.npz`` image of a parsed file (valid for exactly the text file it was written for: size + mtime_ns)
  next to it; ``Dynspec.load_file`` uses it when it is at least as new as the text file;
* :func:`write` emits the text format, one ``str.join`` per sub-integration.
"""
import os

import numpy as np

COLUMNS = "isub ichan time(min) freq(MHz) flux flux_err"
SIDECAR_SUFFIX = ".scint.npz"


def read_table(path):
    """Return (header lines without the leading '#', float64 table [nrow, ncol])."""
    with open(path, "rb") as fh:
        raw = fh.read()
    header, ncol, tokens = [], 0, []
    for line in raw.splitlines():
        s = line.strip()
        if not s:
            continue
        if s.startswith(b"#"):
            header.append(s[1:].strip().decode(errors="replace"))
            continue
        parts = s.split()
        if ncol == 0:
            ncol = len(parts)
        elif len(parts) != ncol:
            raise ValueError(f"{path}: a data line has {len(parts)} columns, expected {ncol}")
        tokens.extend(parts)
    if ncol < 5:
        raise ValueError(f"{path}: not a psrflux file (need at least 5 data columns)")
    table = np.array(tokens, dtype=np.float64).reshape(-1, ncol)
    return header, table


def header_mjd(header):
    """Value of the first 'MJD0:' header line, or None."""
    for line in header:
        words = line.split()
        if words and words[0] == "MJD0:":
            return float(words[1])
    return None


def observation(header, table, mjd=None, mjd_known=None):
    """Attribute dictionary of one observation from a parsed table.

    Definitions follow the reference loader so that the same file gives the same object:
    times in seconds from the first sub-integration, ``mjd`` advanced to that start (unless
    given), ``df`` / ``bw`` / ``freq`` rounded to 1e-5 / 1e-2 / 1e-2 MHz, the flux plane as
    ``dyn[nchan, nsub]`` with ascending frequency."""
    isub, ichan, t_min, f_mhz, flux = (table[:, k] for k in range(5))
    times = np.unique(t_min * 60)
    start = mjd_known if mjd_known is not None else header_mjd(header)
    if mjd is None:
        if start is None:
            raise ValueError("psrflux file has no 'MJD0:' header line and no mjd was given")
        mjd = start + times[0] / 86400
    times = times - times[0]
    nchan = int(ichan.max()) + 1
    nsub = int(isub.max()) + 1
    span = f_mhz[-1] - f_mhz[0]              # signed: negative when channels descend in the file
    df = round(span / nchan, 5)
    bw = round(span + df, 2)
    dt = np.mean(np.diff(times))
    freqs = np.unique(f_mhz)
    dyn = flux.reshape(nsub, nchan).T
    if df < 0:
        df, bw, dyn = -df, -bw, dyn[::-1]
    return dict(header=header, times=times, mjd=mjd, nchan=nchan, nsub=nsub, df=df, bw=bw, dt=dt,
                tobs=times.max() + dt, freqs=freqs, freq=round(np.mean(freqs), 2),
                dyn=np.ascontiguousarray(dyn))


def leading_short_subs(times, threshold):
    """Number of sub-integrations to drop at the start: the first interval is compared with
    the mean and scatter of the remaining ones until it is no longer short by more than
    `threshold` standard deviations."""
    gaps = np.abs(np.diff(times))
    k = 0
    while True:
        rest = gaps[k + 1:]
        mean, sdev = np.mean(rest), np.std(rest)
        if not (gaps[k] - mean <= -threshold * sdev and sdev >= 0):
            return k
        k += 1


def write(path, header, mjd, times, freqs, dyn, note=None):
    """Write `dyn[nchan, nsub]` in psrflux format (flux_err column 0, time in minutes)."""
    out = ["# Scintools-modified dynamic spectrum in psrflux format\n",
           "# Created using write_file method in Dynspec class\n"]
    if note is not None:
        out.append(f"# Note: {note}\n")
    out.append(f"# MJD0: {mjd}\n")
    out.append("# Original header begins below:\n")
    out.extend(f"# {line} \n" for line in header)
    if not any("isub" in line for line in header):
        out.append(f"# {COLUMNS}\n")
    chan = range(len(freqs))
    with open(path, "w") as fh:
        fh.writelines(out)
        for i, t in enumerate(times):
            minute = t / 60
            col = dyn[:, i]
            fh.write("".join(f"{i} {j} {minute} {freqs[j]} {col[j]} 0\n" for j in chan))


def sidecar_path(path):
    return path + SIDECAR_SUFFIX


def _stamp(path):
    """Identity of the text file a side-car belongs to: size and modification time to the nanosecond."""
    st = os.stat(path)
    return np.array([st.st_size, st.st_mtime_ns], dtype=np.int64)


def save_sidecar(path, header, table):
    """Binary image of a parsed text file (loaded instead of re-parsing while it is current).  The text
    file's size and mtime_ns are stored with it: a replaced text file -- even one with an OLDER timestamp
    (cp -p, rsync, untar) -- never matches."""
    np.savez(sidecar_path(path), header=np.array(header, dtype=str), table=table, stamp=_stamp(path))


def load_sidecar(path):
    """(header, table) from the side-car if it exists and was written for exactly this text file (same size,
    same mtime_ns); None otherwise."""
    side = sidecar_path(path)
    try:
        with np.load(side) as z:
            if not np.array_equal(z["stamp"], _stamp(path)):
                return None
            return [str(s) for s in z["header"]], z["table"]
    except (OSError, KeyError, ValueError):
        return None
