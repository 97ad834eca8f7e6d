from .Blocks import FCN, CRB3d, CRB2d, DeCRB2d  # noqa: F401
