"""CPU oracle for the Mammo-CLIP contrastive pre-training hot path.

TEST INFRASTRUCTURE ONLY.  This package is a plain torch-fp32 (CPU) restatement of the
reference's algorithm for the one path this repo accelerates (EfficientNet-B2/B5 image encoder,
BioClinicalBERT text encoder, linear projection heads, all-gather + symmetric InfoNCE).  It is the
*checker*: only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may
import it.  Nothing under ``mammo_clip_amd/`` (the product) imports it, and the product has no CPU
fallback: it raises if the HIP library is missing.

Parity pin: every function here is checked against golden vectors produced by importing the
reference itself in the build container (``tests/golden/make_golden.py`` -> ``tests/golden/*.npz``,
checked by ``tests/test_oracle_golden.py``).  The reference ships no tests or golden vectors of its own
(SURVEY.md section 4), and the BERT arithmetic lives in the third-party ``transformers`` package
(reference pins 4.41.1, environment.yml:208; this container has 5.15.0), so the pins are the
reference's own outputs generated here.

Each function cites the reference file:line (relative to /root/reference/src/codebase/breastclip)
that it restates.
"""
from . import arch, efficientnet, bert, clip, loss, weights, inputs  # noqa: F401
