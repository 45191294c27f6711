"""CPU oracle for the Reformer-TTS training hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``oracle/`` is part of the product:
only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of
``bench.py`` may import it, and there only as the checker / the timed CPU
baseline -- never as the thing shipped.  The product path
(``reformer-tts_amd/``) raises if its HIP library is missing; it never falls
back to this code.

Parity status:
  * every row except the LSH attention arithmetic is pinned by golden vectors
    produced here by importing the reference's own modules
    (``tests/golden/make_golden.py``);
  * the LSH attention arithmetic lives in the third-party package
    ``reformer-pytorch==0.19.1`` (reference ``requirements.txt:11``), which is
    absent from ``/root/reference`` and from this image: that piece is
    **parity unpinned**.  ``lsh_ref.py`` restates its published algorithm; its
    integer stages (hash / stable sort) are additionally cross-checked against
    HuggingFace's independent implementation of the same paper.
"""
