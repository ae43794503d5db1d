"""TEST INFRASTRUCTURE ONLY — CPU oracle for the RAG Matching-Net forward path.

Nothing under ``oracle/`` is product code.  Only ``tests/``,
``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py`` may
import it, and only as the checker.  The product path (``rag_amd``) must never
import this package; it fails loudly when the HIP library is missing instead.
"""
