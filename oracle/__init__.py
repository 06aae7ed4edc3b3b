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
* TensorFlow-1 half, translational graphs -- FORWARD GRAPH PINNED, optimiser arithmetic unpinned: TF1 is not
  installable here, but ``tests/golden/make_tf_graph_golden.py`` installs a lazily evaluated numpy stand-in for the few
  dozen TF ops involved (``tests/golden/tf_shim.py``) and lets the REFERENCE's own code build its graphs --
  ``_define_variables`` / ``_define_embed_graph`` / ``_define_alignment_graph`` / ``_define_mapping_graph`` of
  basic_model.py, mtranse.py, aligne.py, bootea.py, bootea_transh.py, bootea_rotate.py and models/trans/{transe,transh,
  transd}.py with losses.py, initializers.py, mapping.py underneath.  The loss of a fixed batch and its finite-difference
  gradient w.r.t. every variable are stored (``tests/golden/tf_graphs.npz``); the step functions here reproduce both
  (``tests/test_oracle_golden.py::*reference_graph``).  What stays an assumption is the meaning of the individual TF ops
  (l2_normalize = x * rsqrt(max(sum x^2, 1e-12)), gather gradients summed per row, relu'(0) = 0) and the optimisers'
  arithmetic (Adagrad accumulator 0.1 and no epsilon, TF-Adam's epsilon-hat and its dense updates), as written down in
  DESIGN.md (H1/H3/H4).
* TensorFlow-1 half, GNN graphs -- FORWARD GRAPHS PINNED the same way: ``GCN_Align_Unit`` (structure and attribute unit),
  RDGCN's ``Layer.build()`` and AliNet's ``_generate_rel_graph`` are built by the reference's own code under the
  stand-in; ``gcn_se_epoch`` here, and on the GPU the device models themselves, reproduce outputs, loss and every
  variable's finite-difference gradient (``tests/test_graph_golden.py``).  Assumptions that remain: the single ops
  (``sparse_tensor_dense_matmul``, ``sparse_softmax`` grouping runs of equal rows, keras ``BatchNormalization`` in
  inference mode with epsilon 1e-3, ``conv1d`` with kernel size 1 = a dense layer) and Adam's arithmetic.
* Random draws (negative triples, negative links): the reference uses python ``random``; the
  restatements here define the Philox / keyed-permutation formulation the device kernels
  reproduce bit for bit, and the tests check the reference's invariants on them
  (distinctness, membership, set semantics, exclusion of true triples / seed links) and, for the triple sampler, agreement
  in distribution with 300 runs of the reference's own function (``tests/golden/neg_stats.npz``).
* Host-side matchings (``galeshapley``, ``stable_alignment``): literal restatements of
  ``modules/finding/alignment.py:87-221``.
"""
