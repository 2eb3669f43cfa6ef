% illegal.gml
%
% This test makes sure that the program catches the subscript
% error and returns with an error status.  We rely on this
% property when testing other features (see features.gml)
%

[] -1 get
render
