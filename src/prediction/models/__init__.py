from pkgutil import extend_path

__path__ = extend_path(__path__, __name__)  # fall through to the reference's modules of this package (src/__init__.py)

