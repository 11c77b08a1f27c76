"""CPU oracle for the EEG -> LSTM -> distillation hot path.

TEST INFRASTRUCTURE ONLY.  This package is a plain numpy restatement of the
arithmetic the reference performs through scipy / torch on the path named by
BASELINE.json:north_star.  Only ``tests/``, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of ``bench.py`` may import it; the product
(``cerebralsignalnetworks_amd``) never does and fails loudly when its HIP
library is missing.

Pinning: the oracle is pinned to the reference ITSELF.  The reference has no tests or fixtures of its own
 and most of its modules cannot be imported here (faiss / torchvision / cv2 / librosa /
models.lstm at module level), but the class and function bodies on the hot path need only torch / numpy /
scipy: ``tests/golden/ref_lift.py`` compiles exactly those definitions out of the files where they lie under
/root/reference (nothing copied, nothing stubbed) and ``tests/golden/make_ref_goldens.py`` executes them on
seeded inputs in the build container; the outputs are committed as ``tests/golden/ref_*.npz`` and every
oracle function is checked against them in ``tests/test_ref_pinned.py`` (DESIGN.md section 5 lists fixture ->
executed definitions).  A second, older set of goldens (``make_goldens.py``) comes from the third-party calls
the reference makes (scipy.signal, torch.nn.LSTM, ...) and from the reference modules that do import.
Not executable here and therefore restated only -- "parity unpinned": the faiss search inside ``evaluate``
(its bookkeeping follows the reference text line by line).
"""
