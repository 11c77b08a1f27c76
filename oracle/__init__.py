"""CPU oracle for the EEG -> LSTM -> distillation hot path.

TEST INFRASTRUCTURE ONLY.  This package is a plain numpy restatement of the
arithmetic the reference performs through scipy / torch on the path named by
BASELINE.json:north_star.  Only ``tests/``, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of ``bench.py`` may import it; the product
(``cerebralsignalnetworks_amd``) never does and fails loudly when its HIP
library is missing.

Pinning: the reference has no tests, golden vectors or fixtures of its own
(SURVEY.md section 4), and most of its modules cannot be imported here
(faiss / torchvision / models.lstm are absent).  The oracle is therefore pinned
against outputs of the third-party calls the reference itself makes
(``scipy.signal.butter/sosfilt/filtfilt``, ``torch.nn.LSTM`` on CPU,
``torch.nn.CosineSimilarity``, ``F.cross_entropy``, ``nn.KLDivLoss``) and of
the reference pieces that do import (``utils/EEGFilters.py`` band edges,
``EEG-BarlowNetworks/optim.py`` LARS, ``utils/utils.py`` cosine_scheduler).
Those outputs are committed under ``tests/golden/`` with the script that made
them (``tests/golden/make_goldens.py``).
"""
