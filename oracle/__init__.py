"""CPU fp32 ORACLE for the Text-To-Video-Finetuning denoising train step.

TEST INFRASTRUCTURE ONLY.  Nothing in the product package
(`text-to-video-finetuning_amd/`, imported as `t2v_amd`) may import this
package.  Only `tests/`, `__graft_entry__.smoke()` and the `cpu_baseline` leg
of `bench.py` use it, and only as the checker / the timed CPU baseline.

PARITY UNPINNED: the reference repository holds no tests, no
golden vectors and no known-answer fixtures for this path, and the arithmetic
of its leaf operators lives in an un-vendored, un-pinned third-party
dependency (`diffusers`, requirements.txt:5 of the reference, era v0.17-0.18)
that is not installable here.  This package restates that published algorithm
in plain PyTorch fp32, following the reference's own wiring files line by line
(`models/unet_3d_condition.py`, `models/unet_3d_blocks.py`, `train.py`), and is
pinned only where the reference is executable in the build container:
`utils/lora.py` (LoRA layers + injection), `utils/bucketing.py` and the
state-dict key schema of `utils/convert_diffusers_to_original_ms_text_to_video.py`
— see `tests/golden/make_golden.py` for the generating script and
`tests/test_oracle_*.py` for the checks.
"""
