"""oracle/ -- CPU restatement of the OpenEA hot path.  TEST INFRASTRUCTURE ONLY.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may
import this package, and only as the checker / reported baseline -- never as the thing
measured or shipped.  Nothing under ``openea_amd/`` imports it.

Pinning (details in DESIGN.md, section "Oracle"):

* numpy half of the reference (positive batching, neighbour search, similarity, CSLS,
  greedy alignment / rank, early stop) -- PINNED: ``tests/golden/make_golden.py`` imports
  the reference's own modules in place (``/root/reference``, this container only) and
  stores their outputs as small fixtures; ``tests/test_oracle_golden.py`` checks every
  restatement here against them.
* TensorFlow-1 half (l2_normalize, translational losses, gradient through the gather,
  Adagrad/SGD/Adam, sparse_tensor_dense_matmul, sparse_softmax, BatchNormalization) --
  PARITY UNPINNED: TF1 is not installable here and the reference ships no golden vectors.
  The restatements follow the cited reference lines plus the TF1 op semantics written down
  in DESIGN.md (assumptions H1/H3/H4).  What can be pinned without TF is: every hand-derived
  gradient (translational step for all losses, TransH, GCN-Align epoch, sparse attention) is
  checked against finite differences of the loss in ``tests/test_oracle_golden.py``.
* Random draws (negative triples, negative links): the reference uses python ``random``; the
  restatements here define the Philox / keyed-permutation formulation the device kernels
  reproduce bit for bit, and the tests check the reference's invariants on them
  (distinctness, membership, set semantics, exclusion of true triples / seed links).
* Host-side matchings (``galeshapley``, ``stable_alignment``): literal restatements of
  ``modules/finding/alignment.py:87-221``.
"""
