"""Drop-in model package: `t2v_amd.models.unet_3d_condition.UNet3DConditionModel` mirrors the reference's
`models.unet_3d_condition.UNet3DConditionModel` (models/unet_3d_condition.py:53)."""
