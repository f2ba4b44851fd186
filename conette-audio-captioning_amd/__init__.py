"""conette_amd -- MI355X-native CoNeTTE inference path (drop-in for the reference's
``CoNeTTEConfig`` / ``CoNeTTEModel`` API)."""
