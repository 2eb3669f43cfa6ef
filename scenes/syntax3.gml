% syntax3.gml
%
% unbound variable reference.
%

x render

