"""Model registry: ``config.network`` -> constructor call, as the reference's if/elif chain
(train.py:324-373, predict.py:233-276).  Networks on the hot path are built from the
MI355X drop-ins; the reference's other networks are out of scope and named as such."""

IN_SCOPE = ("unet", "vnet", "res_unet", "unetr", "IS", "csrnet", "re_net", "er_net")
OUT_OF_SCOPE = ("densenet", "vtnet", "densevoxelnet", "dunet")


def build_model(config):
    """``config`` needs .network, .in_classes, .out_classes (attribute or key access)."""
    get = (lambda k: config[k]) if isinstance(config, dict) else (lambda k: getattr(config, k))
    network = get("network")
    if network == "unet":                                   # train.py:328-331
        from .models.three_d.unet3d import UNet3D
        return UNet3D(in_channels=get("in_classes"), out_channels=get("out_classes"), init_features=32)
    if network == "vnet":                                   # train.py:358-361
        from .models.three_d.vnet3d import VNet
        return VNet(in_channels=get("in_classes"), classes=get("out_classes"))
    if network == "res_unet":                               # train.py:324-327
        from .models.three_d.residual_unet3d import UNet
        return UNet(in_channels=get("in_classes"), n_classes=get("out_classes"), base_n_filter=32)
    if network == "unetr":                                  # train.py:346-349 (the reference ignores in/out classes)
        from .models.three_d.unetr import UNETR
        return UNETR()
    if network == "IS":                                     # train.py:340-343
        from .models.three_d.IS import UNet3D as ISNet
        return ISNet(in_channels=get("in_classes"), out_channels=get("out_classes"), init_features=32)
    if network == "csrnet":                                 # train.py:362-365
        from .models.three_d.csrnet import CSRNet
        return CSRNet(in_channels=get("in_classes"), out_channels=get("out_classes"))
    if network == "re_net":                                 # train.py:336-339
        from .models.three_d.RE_net import RE_Net
        return RE_Net()
    if network == "er_net":                                 # train.py:332-335
        from .models.three_d.ER_net import ER_Net
        return ER_Net(classes=get("out_classes"), channels=get("in_classes"))
    if network in OUT_OF_SCOPE:
        raise NotImplementedError(f"network '{network}' is outside the MI355X hot-path scope (SURVEY.md section 2)")
    raise ValueError(f"unknown network '{network}'")
