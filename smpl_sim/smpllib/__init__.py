"""Shim package: the motion library modules here shadow the reference's; its other smpllib modules resolve behind them."""
from pkgutil import extend_path
__path__ = extend_path(__path__, __name__)
