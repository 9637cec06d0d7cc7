"""CPU oracle for the DDPM / UNetv2 hot path.

TEST INFRASTRUCTURE ONLY.  Nothing in the product package (``rho_diffusion_amd``)
may import this package; only ``tests/``, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of ``bench.py`` use it, and only as the checker / baseline.

``oracle.ref_torch`` is a plain fp32 restatement (stock PyTorch CPU ops) of the
reference algorithm (intel/rho-diffusion @ 2024_10_08); every function cites the
reference ``file:line`` it follows.  It is pinned by golden vectors generated from
the reference itself in the build container (``tests/golden/make_golden.py`` ->
``tests/golden/*.npz``) and by the reference's only known-answer test
(``tests/pipeline/test_schedule.py:28-46``); see ``tests/test_oracle_golden.py``.

Pinned: schedules, embeddings, UNetv2 modules / forward / gradients, DDPM loops, AdamW (g1-g8), the
GaussianDiffusionPipeline sampling path (g9) and training_step (g11), ExponentialMovingAverage (g10).
PARITY UNPINNED: the ``dds_*`` functions (diffusers-style DDPMScheduler / training loss of ``DiffusersDDPMPipeline``): that arithmetic
lives in the third-party ``diffusers`` package (unpinned in the reference's pyproject, not installed here); they restate
the published scheduler and are checked for structure only.
"""
