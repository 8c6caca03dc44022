from .dgp import GCN_Dense_Att, GraphConv, GraphOperator, fold_groups, group_edges

__all__ = ["GCN_Dense_Att", "GraphConv", "GraphOperator", "fold_groups", "group_edges"]
