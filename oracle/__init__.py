"""CPU oracle for the LitePose inference hot path.  TEST INFRASTRUCTURE ONLY.

Everything under ``oracle/`` is a CPU restatement of the reference algorithm
(mit-han-lab/litepose) used as the *checker* for the HIP path.  Only
``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg
may import it.  The product package ``litepose_amd`` never does: it fails loudly
when the HIP library is missing.

Pinning status: PINNED.  The reference publishes no golden vectors (SURVEY.md
§4), so every module here is pinned against outputs of the reference's own
Python files imported from /root/reference in the build container
(``tests/golden/gen_golden.py`` is the generating script; the fixtures it wrote
are committed under ``tests/golden/``).

Modules
  spec.py        arch-JSON -> layer list + reference state_dict key scheme
                 (lib/models/pose_mobilenet.py:22-135, lib/models/layers/layers.py)
  net_ref.py     fp32 torch-functional restatement of LitePose.forward
                 (pose_mobilenet.py:137-156, layers.py:18-24,90-133)
  inference_ref.py  flip-TTA / multi-stage merge / projection
                 (lib/core/inference.py:75-173,176-208; valid.py:224-225)
  munkres_ref.py restatement of munkres 1.1.4 (third-party, un-vendored;
                 requirements.txt:12 unpinned; call site lib/core/group.py:19-23)
  group_ref.py   NumPy restatement of HeatmapParser (lib/core/group.py:26-291)
  transforms_ref.py  get_final_preds / get_multi_scale_size
                 (lib/utils/transforms.py:50-99,155-176,195-202)
  synth.py       seeded synthetic weights / images / AE blob maps
"""
