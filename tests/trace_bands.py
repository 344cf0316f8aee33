"""Relative bands for the generator-side scalars of the 32-iteration trajectory test (tests/test_graph_gpu.py).

The trajectory is a GAN: after D's first Adam step every discriminator weight has moved by +-lr, so everything that
passes through D is chaotic and is NOT compared beyond iteration 0.  The segmentor-side scalars are compared along the
whole trajectory; how tightly is bounded by how far the REFERENCE's own arithmetic moves when only its rounding changes
(tests/golden/iter_trace.npz vs iter_trace_f64.npz: the same replay of the reference modules in fp32 and fp64).
``tests/test_oracle_golden.py::test_trace_bands_cover_reference_fp_spread`` asserts band >= measured spread and
band <= 4x spread + 1e-2, so the bands can neither be tighter than the reference itself nor drift arbitrarily wide."""
# measured spread of the reference (max over 32 iterations of |fp32 - fp64| / |fp64|):
#   G_seg 3.4e-3, G_semi 2.3e-2, G_rec 0.53, G_nce 0.18     (G_rec / G_nce pass through the translator, which D trains)
# the HIP path against the fp32 reference trace on the same draws: 3.2e-3, 2.1e-2, 0.30, 0.31-0.63 (two builds whose only
# difference is the order of a few fp32 sums land on G_nce trajectories that far apart: it is the most D-coupled of the four)
TRACE_BANDS = {"G_seg": 0.01, "G_semi": 0.05, "G_rec": 0.60, "G_nce": 0.70}
