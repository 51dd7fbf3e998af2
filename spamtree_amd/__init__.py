"""spamtree_amd: MI355X-native per-Gibbs-sweep DAG-node linear algebra for spamtree."""
__version__ = "0.1.0"
