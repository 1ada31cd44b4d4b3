"""`models` package of the reference (codes/models/__init__.py): create_model(opt) -> model wrapper."""


def create_model(opt, *kargs, **kwargs):
    model = opt['model']
    if model in ('srragan', 'srgan'):
        # 'srgan' (codes/models/SRGAN_model.py) cannot run in the reference (undefined `need_HR`, :127-130); both names
        # resolve to the live SRRaGAN wrapper here.
        from .SRRaGAN_model import SRRaGANModel as M
    elif model == 'dncnn':
        raise NotImplementedError('Model [dncnn] (explorable JPEG decoding) is outside the RRDB+CEM hot path')
    else:
        raise NotImplementedError('Model [{:s}] not recognized.'.format(model))
    m = M(opt, *kargs, **kwargs)
    print('Model [{:s}] is created.'.format(m.__class__.__name__))
    return m
