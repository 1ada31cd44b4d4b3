"""`models` package of the reference (codes/models/__init__.py): create_model(opt) -> model wrapper."""
import importlib

# opt['model'] -> (module, class).  'srgan' (codes/models/SRGAN_model.py) cannot run in the reference (undefined `need_HR`, :127-130); both
# GAN names resolve to the live SRRaGAN wrapper here.  'dncnn' (explorable JPEG decoding) is outside the RRDB+CEM hot path.
_WRAPPERS = {'srragan': ('SRRaGAN_model', 'SRRaGANModel'), 'srgan': ('SRRaGAN_model', 'SRRaGANModel')}
_OUT_OF_SCOPE = {'dncnn': 'Model [dncnn] (explorable JPEG decoding) is outside the RRDB+CEM hot path'}


def create_model(opt, *kargs, **kwargs):
    name = opt['model']
    if name in _OUT_OF_SCOPE:
        raise NotImplementedError(_OUT_OF_SCOPE[name])
    if name not in _WRAPPERS:
        raise NotImplementedError('Model [{:s}] not recognized.'.format(name))
    module, cls = _WRAPPERS[name]
    wrapper = getattr(importlib.import_module('.' + module, __name__), cls)(opt, *kargs, **kwargs)
    print('Model [{:s}] is created.'.format(type(wrapper).__name__))
    return wrapper
