"""t2v_amd — MI355X-native denoising-step path behind the Text-To-Video-Finetuning module API.

Import as `t2v_amd` (the directory name `text-to-video-finetuning_amd` is not a Python identifier;
`/root/repo/t2v_amd.py` registers this directory under that name).
"""
__version__ = "0.1.0"
