"""Import shim: lets the reference's drivers (``from src.fm import
FactorizationMachines``) resolve to the MI355X implementation when this
repository precedes the reference on ``sys.path``.  See INTEGRATION.md."""
