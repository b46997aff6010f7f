"""oracle/ -- CPU restatement of FrameINO's denoising hot path.  TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's `cpu_baseline` leg may import this
package, and there only as the checker / reported baseline.  Nothing under frameino_amd/
imports it; the product path raises if the HIP library is missing.

What it is: a plain-PyTorch (CPU, any dtype; fp32 is the reference precision) functional
restatement of the reference's algorithm, operating on flat state-dicts keyed by the
reference's own parameter names.  Each function cites the reference file:line it follows.

How it is pinned: tools/golden/make_golden.py imports the reference's own model files from
/root/reference (through the builder-written `diffusers` stand-in under
tools/golden/diffusers_stub -- diffusers itself is not installable here), runs them on
seeded inputs with seeded random weights, and commits inputs+weights+outputs under
tests/golden/*.npz.  tests/test_oracle_golden.py checks every oracle function against
those vectors.  Third-party arithmetic that is NOT in /root/reference (diffusers'
FeedForward / RMSNorm / FP32LayerNorm / LayerNormZero / AdaLayerNorm / schedulers) is
restated from its published semantics: for those pieces parity is UNPINNED.
"""
