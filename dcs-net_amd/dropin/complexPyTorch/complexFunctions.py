from dcsnet.complexFunctions import complex_upsample, complex_relu  # noqa: F401
