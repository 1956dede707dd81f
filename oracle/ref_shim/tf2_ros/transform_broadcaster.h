// included by the node sources, never used
