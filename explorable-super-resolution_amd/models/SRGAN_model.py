"""The reference's legacy 'srgan' wrapper (codes/models/SRGAN_model.py) cannot run there (undefined `need_HR`, :127-130); the
name is kept importable and resolves to the live SRRaGAN wrapper."""
from .SRRaGAN_model import SRRaGANModel as SRGANModel  # noqa: F401
