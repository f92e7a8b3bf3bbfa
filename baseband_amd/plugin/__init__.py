"""The ``baseband.io`` plugin modules (pyproject.toml, entry-point group
``baseband.io``): one module per format with ``open`` and ``info``, as
/root/reference/baseband/io/__init__.py:162-231 requires of a format, whose
stream readers and writers answer with the reference's types -- `Time`,
`Quantity`, NumPy arrays (`_proxy.ReferenceTyped`)."""
