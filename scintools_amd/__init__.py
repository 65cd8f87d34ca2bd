"""scintools_amd -- MI355X-native secondary-spectrum + theta-theta hot path.

Drop-in for ``scintools.dynspec.Dynspec.calc_sspec`` and the theta-theta
functions of ``scintools.ththmod`` (thth_map, thth_redmap, rev_map, modeler,
chisq_calc, Eval_calc, single_search), computed by hand-written HIP kernels for
gfx950 behind a C ABI (include/scint_hip.h).  Importing the package does not
touch the GPU; the first compute call does, and fails loudly without one.
"""
__version__ = "0.1.0"
