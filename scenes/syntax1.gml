% syntax1.gml
%
% bad array syntax

[ [ { ] } ]

